set -e
export PYTHONUNBUFFERED=1
echo "fused"; timeout -k 10 300 python tools/small_latency.py
echo "three launches"; BOSS_FEW_FUSED=0 timeout -k 10 300 python tools/small_latency.py
