mkdir -p gpurun_out/prof3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS_ATOMIC SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU"; do
i=$((i+1))
rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/prof3/p$i -- python bench.py --iters 10 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof3/p$i.log 2>&1
echo "set $i exit=$?"
done
