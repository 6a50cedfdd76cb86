#!/bin/bash
# round 3: fabric-side traffic of the fused top-k launches with the eight-group split (on the GPU box)
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r03_pmc}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
T=$GRAFT_REPO_ROOT/tools/prof_encode.py
run() {  # tag, args...
  tag=$1; shift
  for pmc in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 150 rocprofv3 --pmc $pmc --kernel-trace -d $O/pmc_${tag}_$pmc -o p --output-format csv -- python3 $T "$@" --launches 6 > $O/pmc_${tag}_$pmc.log 2>&1 || exit 1
    echo "$tag $pmc done"
  done
}
run f32_24576 --d-sae 24576 --tokens 65536 && run f32_49152 --d-sae 49152 --tokens 65536 && \
run f32_131072 --d-sae 131072 --tokens 16384 && run f32_131072_s1 --d-sae 131072 --tokens 16384 --slots 1 && \
run bf16_131072 --bf16 --d-sae 131072 --tokens 16384 && run bf16_131072_s1 --bf16 --d-sae 131072 --tokens 16384 --slots 1 || exit 1
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $O/r03_encode_topk_pmc.json \
  24576x65536=$O/pmc_f32_24576_FETCH_SIZE,$O/pmc_f32_24576_WRITE_SIZE \
  49152x65536=$O/pmc_f32_49152_FETCH_SIZE,$O/pmc_f32_49152_WRITE_SIZE \
  131072x16384=$O/pmc_f32_131072_FETCH_SIZE,$O/pmc_f32_131072_WRITE_SIZE \
  131072x16384x1slot=$O/pmc_f32_131072_s1_FETCH_SIZE,$O/pmc_f32_131072_s1_WRITE_SIZE \
  131072x16384xbf16=$O/pmc_bf16_131072_FETCH_SIZE,$O/pmc_bf16_131072_WRITE_SIZE \
  131072x16384xbf16x1slot=$O/pmc_bf16_131072_s1_FETCH_SIZE,$O/pmc_bf16_131072_s1_WRITE_SIZE > $O/pmc_summary.txt
