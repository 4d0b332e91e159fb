import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f))
    print(f)
    for cfg, rows in d['configs'].items():
        for row in rows:
            print(cfg, row['n_ranks'], {k: (round(v['local_ms'], 3), round(v['close_ms'], 3), round(v['comm_ms_ring'], 3)) for k, v in row['by_kind'].items()},
                  'setup', round(row['setup_ms'], 3), 'hybrid', round(row['fronts']['hybrid']['predicted_ms_per_step'], 3))
            print('   ' + ' '.join(f"{p['k']}{p['kind'][0]}:{p['max_local_ms']:.3f}/{p['sum_local_ms']:.2f}" for p in row['passes']))
