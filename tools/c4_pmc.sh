#!/bin/bash
# fabric-side bytes of the config-4 kernels: one stream, one counter set per pass (on the GPU box)
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-c4pmc}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; export TSC_LENS_STREAMS=1
for pmc in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $pmc --kernel-trace -d $O/pmc_$pmc -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/prof_config4.py 2 > $O/pmc_$pmc.log 2>&1 || exit 1
done
