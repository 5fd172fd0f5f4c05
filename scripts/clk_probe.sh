export TMPDIR=/tmp
for m in fast strict; do
  python bench.py --steps 1500 --warmup 3 --mode $m --cpu-seconds 0 > gpurun_out/clk_$m.json 2>/dev/null &
  PID=$!
  sleep 9
  for i in 1 2 3 4; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power|Temperature \(Sensor (edge|junction)" | tr '\n' ' '; echo; sleep 1.5; done
  wait $PID
  python -c "import json; r=json.load(open('gpurun_out/clk_$m.json')); print('$m', r['kernel_ms'], r['value'])"
done
rocm-smi --showmaxpower 2>/dev/null | grep -i power
