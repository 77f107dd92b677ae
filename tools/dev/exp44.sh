export SKYJO_LIB=$PWD/build_exp/coop_sf.so
i=0
for ctrs in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
            "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
            "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum TCC_EA0_RDREQ_DRAM_sum" \
            "TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_SRC_FIFO_FULL_sum" \
            "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_MISS_sum" \
            "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
            "TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum"; do
  i=$((i+1))
  bash tools/gpu_pmc.sh sf_$i "$ctrs" > gpurun_out/exp44_$i.txt 2>&1
  grep -A5 "k_deal" gpurun_out/exp44_$i.txt | grep -v "k_step\|k_reset"
done
