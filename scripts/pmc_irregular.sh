cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03k/pmc_irr
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/a -- python $GRAFT_REPO_ROOT/scripts/time_irregular.py 1e7 matern52 > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/b -- python $GRAFT_REPO_ROOT/scripts/time_irregular.py 1e7 matern52 > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_GDS --kernel-trace --output-format csv -d $OUT/c -- python $GRAFT_REPO_ROOT/scripts/time_irregular.py 1e7 matern52 > $OUT/c.log 2>&1
python $GRAFT_REPO_ROOT/scripts/sq_counters.py $OUT/a $OUT/b $OUT/c --match per-step 2>&1 | head -5
python $GRAFT_REPO_ROOT/scripts/sq_counters.py $OUT/a $OUT/b $OUT/c > $OUT/summary.txt 2>&1
grep -A 30 "tgp_s::k_apply_filter<3, false, 2" $OUT/summary.txt | head -40
