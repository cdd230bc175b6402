# r06an: the complete -m gpu suite with the LDS-DMA forward kernel forced on EVERY shape of its envelope (EDET_PW_GLDS=2; the
# default sends only the SE-gated views there): bit-identical kernels, so every network-level test must pass unchanged.
# The bit-equality test itself switches the variable per call and is unaffected.
mkdir -p gpurun_out
export TMPDIR=/tmp EDET_SKIP_SLOW=1 EDET_PW_GLDS=2
(timeout 2000 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider 2>&1 | cut -c1-2000 | tail -30) > gpurun_out/r06an_pytest_glds2.log
tail -6 gpurun_out/r06an_pytest_glds2.log
