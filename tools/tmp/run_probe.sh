for c in c4 c5; do
PROBE_CFG=$c timeout -k 10 300 python tools/probe_trace.py 2>/dev/null > gpurun_out/probe_${c}_w8.log
done
