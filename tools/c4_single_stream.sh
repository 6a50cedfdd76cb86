#!/bin/bash
# kernel-level view of the config-4 pass: everything on one stream, under rocprofv3 (on the GPU box)
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-c4ss}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; export TSC_LENS_STREAMS=1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof -o c4 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/prof_config4.py 6 > $O/prof.log 2>&1
