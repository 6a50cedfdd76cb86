#!/bin/bash
# Round-2 measurement pass on the GPU box (one gpurun call): kernel stats of the bench, PMC traffic of the
# dominant kernel at the bench's launch shapes and at a footprint beyond the Infinity Cache, evidence for the bf16
# kernel.  Every GPU step has its own time limit; the script stops at the first failure.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2m
mkdir -p $O
step() { echo "== $*"; timeout -k 10 "$@"; }
step 240 rocprofv3 --kernel-trace --stats -d $O/bench_stats -o b --output-format csv -- python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-config4 --no-screened > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
for shape in "24576 65536" "49152 65536" "131072 16384"; do
  set -- $shape
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $pmc | cut -d' ' -f1)
    step 120 rocprofv3 --pmc $pmc --kernel-trace -d $O/pmc_$1_${tag} -o p --output-format csv -- python3 tools/prof_encode.py --d-sae $1 --tokens $2 --launches 6 > $O/pmc_$1_${tag}.log 2>&1
  done
done
python3 tools/pmc_summary.py $O/r02_encode_topk_pmc.json \
  24576x65536=$O/pmc_24576_FETCH_SIZE,$O/pmc_24576_WRITE_SIZE,$O/pmc_24576_SQ_BUSY_CU_CYCLES \
  49152x65536=$O/pmc_49152_FETCH_SIZE,$O/pmc_49152_WRITE_SIZE,$O/pmc_49152_SQ_BUSY_CU_CYCLES \
  131072x16384=$O/pmc_131072_FETCH_SIZE,$O/pmc_131072_WRITE_SIZE,$O/pmc_131072_SQ_BUSY_CU_CYCLES > $O/pmc_summary.txt
# bf16 kernel (config 5): kernel stats and the LDS / MFMA counters
step 120 rocprofv3 --kernel-trace --stats -d $O/bf16_stats -o b --output-format csv -- python3 tools/prof_encode.py --bf16 --d-sae 131072 --tokens 16384 --launches 20 > $O/bf16_stats.log 2>&1
step 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES --kernel-trace -d $O/bf16_pmc -o p --output-format csv -- python3 tools/prof_encode.py --bf16 --d-sae 131072 --tokens 16384 --launches 6 > $O/bf16_pmc.log 2>&1
python3 tools/pmc_summary.py $O/r02_bf16_encode_pmc.json 131072x16384=$O/bf16_pmc > $O/bf16_pmc_summary.txt
step 200 python3 tools/bench_configs.py > $O/configs_4_5.jsonl 2> $O/configs_4_5.err
echo done
