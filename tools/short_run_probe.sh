mkdir -p gpurun_out/short
p() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f M  wall %.2f us  kern %.2f us  info %.1f M' % (d['value']/1e6, d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, (d.get('value_with_info') or 0)/1e6))"; }
for rep in 1 2; do
echo "== default"; python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | p
echo "== HSA_ENABLE_INTERRUPT=0"; HSA_ENABLE_INTERRUPT=0 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | p
done
for g in 1 3 4; do echo "== groups $g"; python bench.py --gpus 1 --steps 20 --warmup 5 --groups $g 2>/dev/null | tail -1 | p; done
echo "== HSA_ENABLE_INTERRUPT=0 groups 4"; HSA_ENABLE_INTERRUPT=0 python bench.py --gpus 1 --steps 20 --warmup 5 --groups 4 2>/dev/null | tail -1 | p
