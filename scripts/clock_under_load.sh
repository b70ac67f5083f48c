python scripts/one_shape.py conv 6000 > gpurun_out/clk_run.log 2>&1 &
PID=$!
sleep 25
for i in 1 2 3; do rocm-smi --showclocks 2>/dev/null | grep -i -E "sclk|mclk|fclk" | head -3; rocm-smi --showpower 2>/dev/null | grep -i -E "power" | head -2; sleep 0.5; done
wait $PID
cat gpurun_out/clk_run.log | tail -1
echo idle; rocm-smi --showclocks 2>/dev/null | grep -i sclk | head -1
