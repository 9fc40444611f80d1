import json,sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], d["ms_per_step"], d["phases_ms"]["rest (all-reduce + scan + init + solve + check + host)"], d["phases_ms"]["untimed_steps_with_events"]["scan+init+solve+check"], repr(d["solver"]["final_loss"]), d["cpu_baseline"].get("gpu_vs_oracle_max_rel_alpha_err"), d["solver"]["iterations"], d["solver"]["loss_evals"])
