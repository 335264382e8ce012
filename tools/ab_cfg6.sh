for v in "" variants/nt0.so; do for o in reference front-to-back; do
  RT_MI355X_LIB=$v timeout -k 10 200 python bench.py --config 6 --steps 32 --warmup 3 --no-cpu-baseline --order $o 2>/dev/null > gpurun_out/c6.json || exit 1
  python3 -c "
import json
d=json.loads(open('gpurun_out/c6.json').read().strip().splitlines()[-1]); r=d['roofline']
print('lib=%s order=%s: %.4f ms per Render, kernel %.4f, nodes/ray %.2f tris/ray %.2f, x16 %.4f, rewalks %s' % ('$v','$o',d['ms_per_step'],r['kernel_avg_ms'],r.get('nodes_per_ray',0),r.get('tris_per_ray',0),d['throughput_mode']['ms_per_frame'], r.get('near_tie_rewalks_per_frame')))"
done; done
