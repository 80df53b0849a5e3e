// loopback_nccl.cpp -- TEST INFRASTRUCTURE: the NCCL / RCCL entry points roger_hip.hip resolves at run time (rccl_api), for N "ranks" that are
// host THREADS of one process on one GPU.  RCCL itself refuses two ranks on one device and a build box has one GPU, so the library's
// code for more than one rank -- the all-reduce of the predicate words in rh_run_steps_dist, route_exchange's grouped send / receive of the
// edge columns with its neighbour arithmetic and buffer offsets -- never ran with nranks > 1 on the device.  With
//     RH_RCCL_LIB=<this library>   (tests/test_hip_loopback_ranks.py builds it with hipcc and starts a child process)
// every rank is a thread with its own rh_ctx, and the collective calls meet here:
//   ncclAllReduce   every rank synchronises its stream, hands in a host copy, the last arrival reduces, every rank copies the result back;
//   ncclSend / Recv inside ncclGroupStart / End: at the end of the group the rank synchronises its stream, posts its sends (device pointers)
//                   to per-pair mailboxes in issue order, performs its receives as device-to-device copies from the peers' posted buffers,
//                   and returns when its own sends have been consumed (the buffers may be reused, as after a completed NCCL group).
// Synchronous where NCCL is stream-ordered: it checks WHAT is exchanged (ranks, offsets, counts, order), not the asynchrony.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace {

struct Msg {
    const void *src;
    size_t bytes;
    bool taken = false, consumed = false;
};
struct World {
    int nranks = 0;
    std::mutex mu;
    std::condition_variable cv;
    // all-reduce rendezvous
    int arrived = 0;
    unsigned long generation = 0;
    std::vector<std::vector<unsigned char>> contrib;
    std::vector<unsigned char> result;
    // mailboxes[src * nranks + dst]
    std::vector<std::deque<std::shared_ptr<Msg>>> box;
    unsigned long sends = 0, recvs = 0, allreduces = 0;
};
struct Comm {
    std::shared_ptr<World> world;
    int rank;
};
struct Op {
    bool send;
    const void *sbuf;
    void *rbuf;
    size_t bytes;
    int peer;
    Comm *comm;
    hipStream_t stream;
};

std::mutex g_mu;
std::map<unsigned long, std::shared_ptr<World>> g_worlds;
unsigned long g_next_id = 1;
thread_local int t_group = 0;
thread_local std::vector<Op> t_ops;

size_t type_size(ncclDataType_t t) {
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

template <class T>
void reduce_into(std::vector<unsigned char> &acc, const std::vector<unsigned char> &x, ncclRedOp_t op) {
    T *a = reinterpret_cast<T *>(acc.data());
    const T *b = reinterpret_cast<const T *>(x.data());
    const size_t n = acc.size() / sizeof(T);
    for (size_t i = 0; i < n; ++i) {
        if (op == ncclMax) a[i] = a[i] > b[i] ? a[i] : b[i];
        else if (op == ncclMin) a[i] = a[i] < b[i] ? a[i] : b[i];
        else a[i] = a[i] + b[i];
    }
}

ncclResult_t run_ops(std::vector<Op> &ops) {
    if (ops.empty()) return ncclSuccess;
    for (const Op &o : ops)
        if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
    World &w = *ops[0].comm->world;
    const int me = ops[0].comm->rank, n = w.nranks;
    std::vector<std::shared_ptr<Msg>> mine;
    {
        std::unique_lock<std::mutex> lk(w.mu);
        for (const Op &o : ops)
            if (o.send) {
                auto m = std::make_shared<Msg>();
                m->src = o.sbuf;
                m->bytes = o.bytes;
                w.box[(size_t)me * n + o.peer].push_back(m);
                mine.push_back(m);
                ++w.sends;
            }
        w.cv.notify_all();
    }
    for (const Op &o : ops) {
        if (o.send) continue;
        std::shared_ptr<Msg> m;
        {
            std::unique_lock<std::mutex> lk(w.mu);
            auto &q = w.box[(size_t)o.peer * n + me];
            w.cv.wait(lk, [&] { return !q.empty(); });
            m = q.front();
            q.pop_front();
            ++w.recvs;
        }
        if (m->bytes != o.bytes) return ncclInvalidArgument;   // a send and its receive disagree on the count
        if (hipMemcpy(o.rbuf, m->src, o.bytes, hipMemcpyDeviceToDevice) != hipSuccess) return ncclUnhandledCudaError;
        {
            std::unique_lock<std::mutex> lk(w.mu);
            m->consumed = true;
            w.cv.notify_all();
        }
    }
    {
        std::unique_lock<std::mutex> lk(w.mu);
        w.cv.wait(lk, [&] {
            for (auto &m : mine)
                if (!m->consumed) return false;
            return true;
        });
    }
    return ncclSuccess;
}

}   // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    std::lock_guard<std::mutex> lk(g_mu);
    std::memset(id, 0, sizeof(*id));
    const unsigned long v = g_next_id++;
    std::memcpy(id->internal, &v, sizeof(v));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    unsigned long v;
    std::memcpy(&v, id.internal, sizeof(v));
    std::shared_ptr<World> w;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto &slot = g_worlds[v];
        if (!slot) {
            slot = std::make_shared<World>();
            slot->nranks = nranks;
            slot->contrib.resize(nranks);
            slot->box.resize((size_t)nranks * nranks);
        }
        if (slot->nranks != nranks) return ncclInvalidArgument;
        w = slot;
    }
    *comm = reinterpret_cast<ncclComm_t>(new Comm{w, rank});
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    delete reinterpret_cast<Comm *>(comm);
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) {
    *count = reinterpret_cast<const Comm *>(comm)->world->nranks;
    return ncclSuccess;
}
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int *rank) {
    *rank = reinterpret_cast<const Comm *>(comm)->rank;
    return ncclSuccess;
}
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "success" : (r == ncclInvalidArgument ? "loopback: invalid argument (mismatched send / receive?)" : "loopback: error"); }

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    Comm *c = reinterpret_cast<Comm *>(comm);
    World &w = *c->world;
    const size_t bytes = count * type_size(type);
    if (!bytes || (type != ncclInt32 && type != ncclFloat64 && type != ncclInt64)) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    std::vector<unsigned char> host(bytes);
    if (hipMemcpy(host.data(), sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    {
        std::unique_lock<std::mutex> lk(w.mu);
        w.contrib[c->rank] = host;
        const unsigned long gen = w.generation;
        if (++w.arrived == w.nranks) {
            w.result = w.contrib[0];
            for (int r = 1; r < w.nranks; ++r) {
                if (w.contrib[r].size() != bytes) return ncclInvalidArgument;
                if (type == ncclInt32) reduce_into<int>(w.result, w.contrib[r], op);
                else if (type == ncclInt64) reduce_into<long long>(w.result, w.contrib[r], op);
                else reduce_into<double>(w.result, w.contrib[r], op);
            }
            w.arrived = 0;
            ++w.generation;
            ++w.allreduces;
            w.cv.notify_all();
        } else {
            w.cv.wait(lk, [&] { return w.generation != gen; });
        }
        host = w.result;   // (stable until every rank has arrived again)
    }
    if (hipMemcpy(recvbuff, host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() {
    ++t_group;
    return ncclSuccess;
}
ncclResult_t ncclGroupEnd() {
    if (--t_group > 0) return ncclSuccess;
    t_group = 0;
    std::vector<Op> ops;
    ops.swap(t_ops);
    return run_ops(ops);
}
static ncclResult_t post(Op o) {
    if (o.peer < 0 || o.peer >= o.comm->world->nranks || o.peer == o.comm->rank) return ncclInvalidArgument;
    t_ops.push_back(o);
    if (t_group > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_ops);
    return run_ops(ops);
}
ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    return post(Op{true, sendbuff, nullptr, count * type_size(type), peer, reinterpret_cast<Comm *>(comm), stream});
}
ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    return post(Op{false, nullptr, recvbuff, count * type_size(type), peer, reinterpret_cast<Comm *>(comm), stream});
}

// what the test asks afterwards: did the exchange really happen?  (totals over every communicator of the process)
void loopback_counts(unsigned long *sends, unsigned long *recvs, unsigned long *allreduces) {
    std::lock_guard<std::mutex> lk(g_mu);
    *sends = *recvs = *allreduces = 0;
    for (auto &kv : g_worlds) {
        std::lock_guard<std::mutex> lw(kv.second->mu);
        *sends += kv.second->sends;
        *recvs += kv.second->recvs;
        *allreduces += kv.second->allreduces;
    }
}
}
