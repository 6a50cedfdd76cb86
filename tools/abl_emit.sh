# ablations of the grouped lens emit at config-4 shape (tools/build_variant.sh builds the libraries); on the GPU box:
#   tools/build_variant.sh noepi "-DTSC_ABL_NOEPI"; tools/build_variant.sh nodma_noepi "-DTSC_ABL_NODMA -DTSC_ABL_NOEPI"
for lib in "" noepi nodma_noepi; do
  if [ -n "$lib" ]; then export TSC_HIP_LIBRARY=$GRAFT_REPO_ROOT/exp/libtsc_$lib.so; else unset TSC_HIP_LIBRARY; fi
  timeout -k 10 120 python tools/prof_emit_group.py 512 10 || exit 1
done
unset TSC_HIP_LIBRARY
timeout -k 10 120 python tools/prof_emit_group.py 512 10 -100 || exit 1   # b_enc = -100: the scan runs, nothing is emitted
