run() { env $1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-dropin --no-inference 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1  %.3f ms/step' % d['ms_per_step'])"; }
for r in 1 2; do
run "TFPP_WGRAD_HALO_WGS=1024"
run "TFPP_WGRAD_HALO_WGS=512"
run "TFPP_WGRAD_HALO_WGS=288"
run "TFPP_WGRAD_HALO_WGS=2048"
done
