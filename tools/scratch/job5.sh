for t in "fuzz_preprocess.py 1200 9001" "fuzz_map_insert.py 500 9002" "fuzz_odometry.py 60 9003" "fuzz_batch.py 250 9004" "fuzz_align.py 300 9005" "fuzz_nn.py 400 9006" "fuzz_bound.py 200 9007" "fuzz_hook_replay.py 100 9008"; do
  set -- $t
  timeout 900 python tools/$1 $2 $3 2>&1 | tail -2 | tr '\n' ' '; echo " <- $t"
done
