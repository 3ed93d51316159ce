"""Summary of a bench.py JSON line (a recording): scripts/rec_summary.py gpurun_out/bench_v6.json"""
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d = json.loads(l)
        print({k: d[k] for k in ('value', 'ms_per_step', 'ms_per_step_median', 'ms_per_step_min')})
        r = d['roofline']
        print('roofline', {k: r.get(k) for k in ('achieved', 'frac', 'executed', 'mfma_issue_frac', 'frac_of_emulation_ceiling', 'traffic', 'ms_per_step', 'launches')})
        g = d['roofline_gae']
        print('gae', {k: g.get(k) for k in ('us_per_launch', 'frac', 'launch_floor_us', 't_over_launch_floor')}, 'saturated', g['saturated'].get('frac'))
        print('mlp', {k: d['roofline_mlp'].get(k) for k in ('achieved', 'frac', 'ms_per_step')})
        print('resident', d['resident_in_hbm'].get('ms_per_step'), {k: round(v['ms_per_step'], 2) for k, v in d['resident_in_hbm'].get('per_seed', {}).items()})
        print('host', (d.get('from_pinned_host') or {}).get('ms_per_step'))
        ri = d['rollout_inference']
        print('rollout', ri['value'], ri['ms_per_call'], 'whole', ri['whole_stack']['value'], ri['whole_stack']['ms_per_call'], ri.get('identical_to_whole_stack'))
        c = d['closed_loop']
        print('closed', c['value'], c['ms_per_iteration'], c['rollout_phase_ms'], c.get('update_alone_ms'))
        sm = d['scaling_model']
        print('scaling', {k: sm.get(k) for k in ('ms_full_update_4096', 'ms_shard_update_512', 'ideal_shard_ms', 'predicted_efficiency_8gpu_compute_only', 'predicted_efficiency_8gpu_compute_only_resident', 'fixed_cost_ms_per_update', 'ms_shard_update_512_with_collectives_world1', 'collectives_enqueue_and_execute_ms_world1', 'predicted_efficiency_8gpu_with_world1_collectives')})
        for k, v in d['configs'].items():
            print(' cfg', k, {kk: v.get(kk) for kk in ('ms_per_update', 'env_steps_per_s', 'resident_ms_per_update')})
        cb = d['cpu_baseline']
        print('cpu', cb['value'], cb['cores'], cb.get('best_thread_count'), cb.get('one_thread', {}).get('value'))
        print('kernel_ms', d.get('kernel_ms_per_step'))
        print('launches', d.get('launches_per_step'))
