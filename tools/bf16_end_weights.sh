for cfg in "10 1234 1" "10 77 1" "25 1234 1" "10 1234 0" "25 1234 0"; do
  set -- $cfg
  TFPP_BN_ROWS=$3 timeout 400 python bench.py --steps $1 --seed $2 --no-cpu-baseline --no-roofline --no-dropin 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); b=d['bf16_vs_autocast_reference']
h=b['hip_bf16_vs_hip_fp32']; c=b['cpu_autocast_vs_cpu_fp32']
print('steps $1 seed $2 BN_ROWS $3 | HIP cos %.4f l2 %.4f med %.4f p99 %.3f | autocast cos %.4f l2 %.4f med %.4f p99 %.3f' % (h['arena_cosine'],h['arena_rel_l2'],h['norm_err_median'],h['norm_err_p99'],c['arena_cosine'],c['arena_rel_l2'],c['norm_err_median'],c['norm_err_p99']))"
done
