"""bench.py's one-line JSON contract (metric, whole-job value, roofline and cpu_baseline objects) on a short run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_line(dev):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1', '--pairs-per-step', '4',
                          '--keypts', '1500', '--cpu-keypts', '64'], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.strip().splitlines() if ln.startswith('{')][-1]
    d = json.loads(line)
    assert d['metric'] == 'registration pairs/sec' and d['unit'] == 'pairs/s' and d['higher_is_better'] is True
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1 and d['scaling'] == 'weak'
    assert d['dtype'] == 'f32' and d['data'] == 'synthetic' and d['vs_baseline'] is None
    assert abs(d['value'] - 4 * 2 / (d['ms_per_step'] * 2e-3)) < 1e-6 * d['value']           # pairs / wall time
    assert 'workload' in d['config'] and 'model' not in d['config']
    r = d['roofline']
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and r['peak'] == 157.3 and r['launches'] == 2
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and 0.3 < r['frac'] < 1.0
    assert r['traffic'] is None or r['traffic'] > 0
    kinds = {o['bound'] for o in d['roofline_other']}
    assert kinds == {'mfma', 'hbm'}
    c = d['cpu_baseline']
    assert c['kind'] in ('reference', 'port') and c['cores'] >= 1 and c['value'] > 0 and c['unit'] == 'pairs/s' and c['sample']
    assert d['config']['registered_ok'].startswith('8/8')
