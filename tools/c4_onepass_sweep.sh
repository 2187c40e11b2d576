for t in "op_l=64 op_w=8" "op_l=64 op_w=4" "op_l=64 op_w=2" "op_l=32 op_w=8" "op_l=32 op_w=4" "op_l=32 op_w=16" "op_l=16 op_w=16" "op_l=16 op_w=8"; do
  set -- $t
  timeout -k 10 120 python bench.py --config C4 --steps 200 --warmup 20 --no-cpu-baseline --tunable $1 --tunable $2 2>/dev/null | python -c "import json,sys; o=json.load(sys.stdin); print('$t', round(o['ms_per_step']*1e3,1), 'us', round(o['hbm_roofline_frac_whole_step'],3), o['assoc_vs_sequential_max_rel_err'])" || echo "$t failed"
done
