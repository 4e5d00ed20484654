timeout -k 10 600 python -m pytest tests/test_gpu_align.py tests/test_gpu_e2e.py -m gpu -x -q 2>&1 | tail -2
timeout -k 10 600 python bench.py > gpurun_out/bench_w8c.json 2> gpurun_out/bench_w8c.err; python - <<XEOF
import json
d=json.loads(open("gpurun_out/bench_w8c.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["ms"], d["roofline"]["frac"])
for k,v in d.get("configs",{}).items(): print(k, v["ms"], v["fasta_identical_to_reference"])
XEOF
