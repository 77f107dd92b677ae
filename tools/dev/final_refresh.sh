bash tools/refresh_profiles.sh > gpurun_out/final_refresh.log 2>&1
bash tools/dev/configs_sweep.sh >> gpurun_out/final_refresh.log 2>&1
SKYJO_LIB= python tools/dev/first60.py >> gpurun_out/final_refresh.log 2>&1
tail -20 gpurun_out/final_refresh.log
