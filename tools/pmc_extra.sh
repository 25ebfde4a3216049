set -e
export TMPDIR=/tmp
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
out=gpurun_out/prof_extra
mkdir -p $out
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format csv -d $out/pmc_lds -- $B > $out/pmc_lds.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_inst -- $B > $out/pmc_inst.log 2>&1
ls $out/*/*/ | head
