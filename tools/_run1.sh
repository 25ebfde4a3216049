mkdir -p gpurun_out/r3 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/run_fused_variants.sh nocov noload noepi nobar none wpe2 base:GPZ_F1_W=96 base:GPZ_F1_W=24 base:GPZ_F1_W=12 > gpurun_out/r3/fused_variants1.txt 2>&1
cat gpurun_out/r3/fused_variants1.txt
B="python3 tools/fused_check.py --time-only"
out=gpurun_out/r3/pmc_f1
mkdir -p $out
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq -- $B > $out/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d $out/pmc_inst -- $B > $out/pmc_inst.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_l2 -- $B > $out/pmc_l2.log 2>&1
python3 tools/pmc_summary.py $out/pmc_sq $out/pmc_inst $out/pmc_l2 --json gpurun_out/r3/pmc_f1.json > gpurun_out/r3/pmc_f1.txt 2>&1
grep -A30 fused_stage1 gpurun_out/r3/pmc_f1.txt | head -40
