"""bayespy_amd — MI355X-native Dirichlet-multinomial maximum-likelihood estimation.

A from-scratch gfx950 implementation of the `dirichletMultinomialEstimation` hot path of
maxsklar/BayesPy behind the reference's own Python surface:

    bayespy_amd.dirichletMultinomialEstimation   drop-in module (same names and signatures)
    bayespy_amd.findDirichletPrior               the CLI (same options, stdin and stdout format)
    bayespy_amd.device                           device-resident statistics / solver objects
    bayespy_amd.parallel                         row-sharded multi-GPU estimation (RCCL)

All computation is in libdme_hip.so (bayespy_amd/csrc, C ABI in include/dme_hip.h).
"""
__version__ = "0.1.0"
