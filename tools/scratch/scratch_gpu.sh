bash profiles/collect.sh r03 > gpurun_out/r03_collect.log 2>&1
tail -60 gpurun_out/r03_collect.log
