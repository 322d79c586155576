OUT=$PWD/gpurun_out/sq255
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $OUT/sqa -o run -- python3 $REPO/tools/shape_probe.py 255x255x255 > $OUT/sqa.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS \
  --kernel-trace --output-format csv -d $OUT/sqb -o run -- python3 $REPO/tools/shape_probe.py 255x255x255 > $OUT/sqb.log 2>&1
cd $REPO
python3 tools/sq_summary.py $OUT/sqa $OUT/sqb > $OUT/sq_summary.md
rm -rf $OUT/sqa $OUT/sqb
