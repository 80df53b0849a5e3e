"""Ready-made setup classes of the hip backend: `svat.SVATSetup`, `oned.ONEDSetup` (forcing held in memory) and
`svat_transport.make_transport_model` (offline transport on top of a SVAT run)."""
