#!/bin/bash
# Profiling artefacts of one round (run on the GPU box through gpurun): rocprofv3 kernel statistics of the default bench.py
# command, the PMC passes MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE in SEPARATE runs, then the SQ counters),
# and a sustained run.  Everything lands under gpurun_out/$1/ ; the summaries are copied to profiles/<round>/ by hand.
#   usage: tools/profile_round.sh r02prof
set -u
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/${1:-prof}
REPO=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$REPO/bench.py"
python3 "$B" --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python3 "$B" --steps 2000 --warmup 20 --no-aux --no-cpu-baseline > "$OUT/bench_sustained_2000.json" 2>/dev/null
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/kt" -o kt -- python3 "$B" --steps 20 --warmup 5 --no-aux --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/kt.err"
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d "$OUT/fetch" -o fetch -- python3 "$B" --steps 2 --warmup 2 --no-aux --no-cpu-baseline > /dev/null 2> "$OUT/fetch.err"
rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d "$OUT/write" -o write -- python3 "$B" --steps 2 --warmup 2 --no-aux --no-cpu-baseline > /dev/null 2> "$OUT/write.err"
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/sq" -o sq -- python3 "$B" --steps 2 --warmup 2 --no-aux --no-cpu-baseline > /dev/null 2> "$OUT/sq.err"
cd "$REPO"
KS=$(find "$OUT/kt" -name '*kernel_stats.csv' | head -1)
FC=$(find "$OUT/fetch" -name '*counter_collection.csv' | head -1)
WC=$(find "$OUT/write" -name '*counter_collection.csv' | head -1)
SC=$(find "$OUT/sq" -name '*counter_collection.csv' | head -1)
cp "$KS" "$OUT/bench_kernel_stats.csv"
python3 tools/summarize_prof.py "$KS" "$FC" "$WC" > "$OUT/summary.md"
python3 tools/make_traffic_json.py "$FC" "$WC" "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 2 --no-aux --no-cpu-baseline; FETCH_SIZE doubled (gfx950)" > "$OUT/traffic.json"
python3 tools/summarize_pmc.py "$SC" > "$OUT/sq_counters_table.md"
# keep the merge small: drop the raw traces
rm -rf "$OUT/kt" "$OUT/fetch" "$OUT/write" "$OUT/sq"
ls -la "$OUT"
