#!/bin/bash
# issue / LDS / L2 counters of the config-4 kernels: one counter set per run (on the GPU box)
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-c4pmc2}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; export TSC_LENS_STREAMS=1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES --kernel-trace -d $O/pmc_sq -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/prof_config4.py 2 > $O/pmc_sq.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum --kernel-trace -d $O/pmc_tcc -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/prof_config4.py 2 > $O/pmc_tcc.log 2>&1 || exit 1
