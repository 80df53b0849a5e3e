#!/usr/bin/env python3
"""Which planes does a step of rh_physics.h need FROM MEMORY, and which does it only produce?

tools/gen_sets.py lists every field a routine mentions; that is what the kernels may load.  This module answers the sharper
question behind `k_step<..., SPARSE>` (rh_run_steps): a field is a PURE OUTPUT of a stage sequence if no execution of the sequence
can observe the value the field had when the sequence began -- every read of it is preceded by an unconditional assignment, and it is
unconditionally assigned before the sequence ends (so the value stored afterwards never is the old one).  Between two steps of ONE
rh_run_steps call nothing but the next step looks at the planes, so the stores of the pure outputs of steps 1 .. n-1 are dead.

The analysis is a conservative flow analysis over the restricted C++ of rh_physics.h (statements, blocks, if / else, for / while,
calls of rt_* / h_* functions that take the column `c`):

    UE   upward-exposed reads: fields that may be read before they are definitely assigned
    DEF  fields definitely assigned on every path
    MAY  fields possibly assigned

    pure outputs of a sequence = DEF - UE          (a field in MAY - DEF keeps its old value on some path: not pure)

Anything the parser does not recognise as a plain `c.x = ...;` statement counts as a read (and an embedded assignment as a possible
write), a call inside `?:`, `&&`, `||` as conditional, a loop body as conditional, and everything behind an early `return` as
conditional.  tests/test_liveness.py checks the analysis on hand-written cases, and the GPU suite poisons the pure outputs before
every step of the golden trajectories."""
import re

ASSIGN_RE = re.compile(r"\bc\.(\w+)\s*(=(?!=)|\+=|-=|\*=|/=)")
MENTION_RE = re.compile(r"\bc\.(\w+)")
CALL_RE = re.compile(r"\b((?:rt|rl|h)_\w+)\s*\(")


class Summary:
    __slots__ = ("ue", "deff", "may", "exits")

    def __init__(self, ue=(), deff=(), may=(), exits=False):
        self.ue, self.deff, self.may, self.exits = set(ue), set(deff), set(may) | set(deff), exits

    def then(self, b):
        """self, then b."""
        ue = self.ue | (b.ue - self.deff)
        deff = set(self.deff) if self.exits else self.deff | b.deff   # behind a possible early return nothing is definite
        return Summary(ue, deff, self.may | b.may, self.exits or b.exits)

    @staticmethod
    def branch(a, b):
        """either a or b (b may be empty: no else)."""
        return Summary(a.ue | b.ue, a.deff & b.deff, a.may | b.may, a.exits or b.exits)

    def conditional(self):
        """may run or not (loop bodies, calls under ?: && ||)."""
        return Summary(self.ue, (), self.may, self.exits)


def _match(text, i, open_ch, close_ch):
    depth = 0
    for j in range(i, len(text)):
        if text[j] == open_ch:
            depth += 1
        elif text[j] == close_ch:
            depth -= 1
            if depth == 0:
                return j
    raise ValueError(f"unbalanced {open_ch}{close_ch} in: {text[i:i + 80]!r}")


def _split_args(s):
    args, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            args.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        args.append(cur.strip())
    return args


class Analyser:
    def __init__(self, funcs):
        """funcs: name -> dict(body=..., refpos=[positions of double&/int& parameters], colpos=[positions of Col& parameters])
        (tools/gen_sets.parse_functions)."""
        self.funcs = funcs
        self.memo = {}

    # -- expressions ---------------------------------------------------------------------------------------------------------------
    def expr(self, text):
        """An expression evaluated once: every mention is a read; embedded assignments are possible writes; calls of known functions
        with the column contribute their summary (sequentially if the expression has no short-circuit / ternary operator)."""
        text = text.strip()
        if not text:
            return Summary()
        reads = set(MENTION_RE.findall(text))
        may = {m.group(1) for m in ASSIGN_RE.finditer(text)}
        out = Summary(ue=reads, may=may)
        unconditional = not re.search(r"\?|&&|\|\|", text)
        for m in CALL_RE.finditer(text):
            callee = m.group(1)
            if callee not in self.funcs:
                continue
            close = _match(text, m.end() - 1, "(", ")")
            args = _split_args(text[m.end(): close])
            f = self.funcs[callee]
            for k in f["refpos"]:      # c.x handed to a double& / int& parameter: read and possibly written
                if k < len(args):
                    fm = re.fullmatch(r"c\.(\w+)", args[k])
                    if fm:
                        out = out.then(Summary(ue={fm.group(1)}, may={fm.group(1)}))
            if any(k < len(args) and args[k] == "c" for k in f["colpos"]):
                cs = self.function(callee)
                out = out.then(cs if unconditional else cs.conditional())
        return out

    def simple(self, stmt):
        stmt = stmt.strip()
        if not stmt or stmt in ("break", "continue"):
            return Summary()
        if stmt == "return" or stmt.startswith("return ") or stmt.startswith("return("):
            s = self.expr(stmt[6:])
            s.exits = True
            return s
        m = re.match(r"^c\.(\w+)\s*(=(?!=)|\+=|-=|\*=|/=)\s*(.*)$", stmt, re.S)
        if m:
            x, op, rhs = m.group(1), m.group(2), m.group(3)
            s = self.expr(rhs)
            if op != "=":
                s = s.then(Summary(ue={x}))
            return s.then(Summary(deff={x}))
        return self.expr(stmt)

    # -- statements ----------------------------------------------------------------------------------------------------------------
    def statement(self, text, i):
        """Parses one statement starting at text[i]; returns (summary, index behind it)."""
        n = len(text)
        while i < n and text[i].isspace():
            i += 1
        if i >= n:
            return Summary(), n
        if text[i] == "{":
            j = _match(text, i, "{", "}")
            return self.block(text[i + 1: j]), j + 1
        m = re.match(r"(if|for|while|else|do|switch)\b", text[i:])
        if m:
            kw = m.group(1)
            if kw in ("do", "switch", "else"):
                raise ValueError(f"liveness: unsupported construct `{kw}` at: {text[i:i + 60]!r}")
            p0 = text.index("(", i)
            p1 = _match(text, p0, "(", ")")
            head = self.expr(text[p0 + 1: p1].replace(";", " , "))
            body, j = self.statement(text, p1 + 1)
            if kw == "if":
                k = j
                while k < n and text[k].isspace():
                    k += 1
                other = Summary()
                if re.match(r"else\b", text[k:]):
                    other, j = self.statement(text, k + 4)
                return head.then(Summary.branch(body, other)), j
            # for / while: the body may not run at all; a value assigned in one iteration and read in the next one is not a
            # read of the incoming value only if it was definitely assigned before the loop -- which `then` accounts for
            return head.then(body.conditional()).then(head.conditional()), j
        # a simple statement up to the next ';' outside parentheses / braces
        depth, j = 0, i
        while j < n:
            ch = text[j]
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
            elif ch == ";" and depth == 0:
                break
            j += 1
        return self.simple(text[i:j]), j + 1

    def block(self, text):
        out, i = Summary(), 0
        while i < len(text):
            s, i = self.statement(text, i)
            out = out.then(s)
        return out

    def function(self, name):
        if name not in self.memo:
            self.memo[name] = Summary()   # (no recursion in rh_physics.h; a cycle would see the empty summary)
            s = self.block(self.funcs[name]["body"])
            s.exits = False                # leaving the callee is not leaving the caller
            self.memo[name] = s
        return self.memo[name]

    def sequence(self, stages):
        out = Summary()
        for rt in stages:
            out = out.then(self.function(rt))
        return out


def pure_outputs(analyser, stages, rotation_pairs=()):
    """Fields a stage sequence only produces: definitely assigned, never read before that, never left holding the incoming value.
    rotation_pairs [(x_m1, x)]: the tau -> taum1 copies `c.x_m1 = c.x` of after_timestep are not reads of x for this purpose when the
    kernel defers the rotation (the copy is made from the x PLANE later, at a point where every plane has been stored)."""
    funcs = analyser.funcs
    if rotation_pairs and "h_rotate" in funcs:
        # analyse with the rotation helper's reads removed: x_m1 = x becomes x_m1 = <nothing>
        body = funcs["h_rotate"]["body"]
        funcs = dict(funcs)
        funcs["h_rotate"] = dict(funcs["h_rotate"], body=re.sub(r"(\bc\.\w+_m1\s*=)\s*c\.\w+\s*;", r"\1 0;", body))
        analyser = Analyser(funcs)
    s = analyser.sequence(stages)
    return (s.deff - s.ue), s
