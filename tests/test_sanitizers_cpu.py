"""SURVEY 5 "race detection / sanitizers", on the CPU builds only (GPU AddressSanitizer is not available on this pool):
  * the oracle (oracle/buffer_oracle.c) and the compiled reference cores behind oracle/ref_shim.cpp are built with
    -fsanitize=address,undefined (make -C oracle asan) and the oracle test files run against those builds in a child
    process with libasan preloaded;
  * the host side of the C ABI (argument validation, workspace arithmetic, the Winograd filter tiling) runs from a
    -fsanitize=undefined build of libbuffer_hip.so (host code only; no GPU call is reached)."""
import ctypes as C
import glob
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _san_lib(name):
    p = subprocess.run(['gcc', f'-print-file-name={name}'], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_and_reference_cores_under_asan_ubsan():
    asan = _san_lib('libasan.so')
    if not asan or not shutil.which('make'):
        pytest.skip('gcc sanitizer runtime not installed')
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'oracle'), 'asan'])
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS='detect_leaks=0:abort_on_error=0:exitcode=86',
               UBSAN_OPTIONS='print_stacktrace=1', BUF_ORACLE_SO=os.path.join(ROOT, 'oracle', 'liboracle_asan.so'))
    ref = os.path.join(ROOT, 'oracle', '_ref', 'libbuffer_ref_asan.so')
    files = [os.path.join(ROOT, 'tests', 'test_oracle_golden.py')]
    if os.path.exists(ref):
        env['BUF_ORACLE_REF_SO'] = ref
        files.append(os.path.join(ROOT, 'tests', 'test_oracle_vs_ref.py'))
    out = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-p', 'no:cacheprovider'] + files, capture_output=True, text=True,
                         env=env, cwd=ROOT, timeout=1500)
    log = out.stdout + out.stderr
    assert out.returncode == 0, log[-4000:]
    assert 'AddressSanitizer' not in log and 'runtime error' not in log, log[-4000:]
    assert ' passed' in out.stdout


_HOST_CASES = r'''
import ctypes as C, sys
import numpy as np
L = C.CDLL(sys.argv[1])
L.buf_last_error.restype = C.c_char_p
for f in ('buf_grid_ws_bytes', 'buf_grid_subsample_ws_bytes', 'buf_fps_ws_bytes', 'buf_select_patches_batched_ws_bytes', 'buf_compact_ws_bytes',
          'buf_knn_ws_bytes', 'buf_segment_instance_norm_ws_bytes'):
    getattr(L, f).restype = C.c_size_t
L.buf_grid_default_cells.restype = C.c_int64
# workspace arithmetic at ordinary and at extreme sizes (no signed overflow, no bad shifts)
for ns, nb in ((0, 1), (1, 1), (30000, 2), (1 << 21, 64), (1 << 24, 1), (2_000_000_000, 1), (2_000_000_000, 65535)):
    cells = L.buf_grid_default_cells(ns, nb)
    assert cells > 0
    assert L.buf_grid_ws_bytes(ns, nb, C.c_int64(cells)) > 0 and L.buf_grid_ws_bytes(ns, nb, C.c_int64(0)) > 0
    assert L.buf_grid_subsample_ws_bytes(ns, nb, C.c_int64(0), 0) > 0 and L.buf_grid_subsample_ws_bytes(ns, nb, C.c_int64(1 << 26), 35) > 0
    assert L.buf_compact_ws_bytes(ns) > 0 and L.buf_select_patches_batched_ws_bytes(ns, nb) > 0
for b, n in ((1, 1), (2, 30000), (64, 1 << 20)):
    assert L.buf_fps_ws_bytes(b, n) >= 0 and L.buf_knn_ws_bytes(b, n, 1) >= 0 and L.buf_segment_instance_norm_ws_bytes(b, 20) > 0
# the host-side Winograd filter tiling, both layouts, and its argument checks
rng = np.random.default_rng(0)
for co, ci in ((32, 16), (64, 48), (128, 64), (128, 128)):
    w = np.ascontiguousarray(rng.standard_normal((co, ci, 3, 3)).astype(np.float32))
    out = np.full(16 * co * ci, np.nan, np.float32)
    assert L.buf_winograd_tile_weights(w.ctypes.data_as(C.c_void_p), co, ci, out.ctypes.data_as(C.c_void_p)) == 0
    assert np.isfinite(out).all() and L.buf_winograd_group(ci, co) in ((2,) if co >= 64 else (1, 2))
assert L.buf_winograd_tile_weights(None, 32, 16, None) == -1 and b'null' in L.buf_last_error()
assert L.buf_winograd_tile_weights(w.ctypes.data_as(C.c_void_p), 30, 16, out.ctypes.data_as(C.c_void_p)) == -1
# argument validation of device entry points: every call returns before the first HIP call
one = C.c_void_p(16)                                          # a non-null pointer that is never dereferenced on the host
assert L.buf_fps(None, 1, 100, 10, None, None, C.c_size_t(0), None) == -1
assert L.buf_vn_gather_block(None, None, None, None, 10, 10, 4, 2, 4, 3, C.c_float(1.0), None, None, None, None, C.c_float(0.2), None, None) == -1
assert L.buf_cylindrical_net_wg(None, 2, None, None, None, None, None, None, None) == -1
ptrs = (C.c_void_p * 8)(*[16] * 8)
relu = (C.c_int * 8)(1, 1, 1, 1, 1, 1, 1, 0)
good_in, good_out = (48, 64, 64, 128, 128, 64, 64, 32), (64, 64, 128, 128, 64, 64, 32, 32)
bad_in = (C.c_int * 8)(40, 64, 64, 128, 128, 64, 64, 32)
assert L.buf_cylindrical_net_wg(one, 2, ptrs, ptrs, bad_in, (C.c_int * 8)(*good_out), relu, one, None) == -1 and b'unsupported widths' in L.buf_last_error()
bad_out = (C.c_int * 8)(64, 64, 128, 128, 64, 64, 32, 64)
assert L.buf_cylindrical_net_wg(one, 2, ptrs, ptrs, (C.c_int * 8)(*good_in), bad_out, relu, one, None) == -1
odd_in, odd_out = (C.c_int * 8)(48, 48, 48, 128, 128, 64, 64, 32), (C.c_int * 8)(48, 48, 128, 128, 64, 64, 32, 32)
assert L.buf_cylindrical_net_wg(one, 2, ptrs, ptrs, odd_in, odd_out, relu, one, None) == -1
assert L.buf_cylindrical_net_wg(one, 0, ptrs, ptrs, (C.c_int * 8)(*good_in), (C.c_int * 8)(*good_out), relu, one, None) == 0      # nothing to do
assert L.buf_version() >= 100
print('host cases ok')
'''


def test_c_abi_host_side_under_ubsan():
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    rts = glob.glob('/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.ubsan_standalone-x86_64.so')
    if not os.path.exists(hipcc) or not rts:
        pytest.skip('hipcc / the clang UBSan runtime are not installed')
    out_so = os.path.join(ROOT, 'build', 'libbuffer_hip_ubsan.so')
    os.makedirs(os.path.dirname(out_so), exist_ok=True)
    src = os.path.join(ROOT, 'buffer_amd', 'csrc', 'buffer_hip.hip')
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O1', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off', '-fno-fast-math',
                           '-Xarch_host', '-fsanitize=undefined', '-Xarch_host', '-fno-sanitize-recover=undefined', '-w', '-o', out_so, src])
    env = dict(os.environ, LD_PRELOAD=rts[0], UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=1')
    out = subprocess.run([sys.executable, '-c', _HOST_CASES, out_so], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    log = out.stdout + out.stderr
    assert out.returncode == 0 and 'host cases ok' in out.stdout, log[-4000:]
    assert 'runtime error' not in log, log[-4000:]
