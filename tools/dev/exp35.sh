# L1 / L2 request counters of the dealing kernel: HEAD build vs the branch-free + peeled build
for v in head peel; do
  export SKYJO_LIB=$PWD/build_exp/$v.so
  bash tools/gpu_pmc.sh ${v}_tcp "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" > gpurun_out/exp35_${v}_tcp.txt 2>&1
  bash tools/gpu_pmc.sh ${v}_tcc "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" > gpurun_out/exp35_${v}_tcc.txt 2>&1
done
grep -A8 "k_deal" gpurun_out/exp35_*_t*.txt | grep -v k_step
