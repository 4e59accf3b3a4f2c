cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/sqbf -o sq -- python3 $R/bench.py --dtype bf16 --steps 2 --warmup 2 --no-aux > /dev/null 2> $R/gpurun_out/sqbf.err
cd $R
python3 tools/summarize_pmc.py $(find gpurun_out/sqbf -name '*counter_collection.csv' | head -1) > gpurun_out/sq_bf16_table.md
rm -rf gpurun_out/sqbf
head -12 gpurun_out/sq_bf16_table.md
