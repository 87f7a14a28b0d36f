"""Builds libapgpu.so (the HIP kernels + C ABI, include/apgpu.h) for gfx950 with hipcc.

``python -m astrophotography_amd._build`` or ``__graft_entry__.build()``.  hipcc cross-compiles without
a GPU; the .so is written next to this file so that it travels with the source tree.
"""
import concurrent.futures as cf
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, 'csrc')
OBJ = os.path.join(CSRC, '_obj')
LIB = os.path.join(PKG, 'libapgpu.so')

SOURCES = ['common.hip', 'elementwise.hip', 'fixbadpix.hip', 'sigclip_global.hip', 'resample.hip', 'resample_stack.hip', 'stack.hip', 'stack_big.hip', 'stack_chunks.hip', 'stack_mad.hip', 'stack_mad_wide.hip', 'stack_mad_pairs.hip', 'stack_mad_pairs_wide.hip', 'combine_f64.hip', 'background.hip', 'lacosmic.hip'] + [
    'stack_inst_f32_calib_h.hip',
    'stack_inst_f32_plain_h.hip',
    'stack_inst_u16_calib_h.hip',
    'stack_inst_u16_plain_h.hip',
    'stack_inst_f32_calib_o.hip',
    'stack_inst_f32_plain_o.hip',
    'stack_inst_u16_calib_o.hip',
    'stack_inst_u16_plain_o.hip',
    'stack_inst_f32_calib_g.hip',
    'stack_inst_f32_plain_g.hip',
    'stack_inst_u16_calib_g.hip',
    'stack_inst_u16_plain_g.hip',
    'stack_inst_f32_calib_j.hip',
    'stack_inst_f32_plain_j.hip',
    'stack_inst_u16_calib_j.hip',
    'stack_inst_u16_plain_j.hip',
    'stack_inst_f32_calib_f.hip',
    'stack_inst_f32_plain_f.hip',
    'stack_inst_u16_calib_f.hip',
    'stack_inst_u16_plain_f.hip',
    'stack_inst_f32_calib_n.hip',
    'stack_inst_f32_plain_n.hip',
    'stack_inst_u16_calib_n.hip',
    'stack_inst_u16_plain_n.hip',
    'stack_inst_f32_calib_e.hip',
    'stack_inst_f32_plain_e.hip',
    'stack_inst_u16_calib_e.hip',
    'stack_inst_u16_plain_e.hip',
    'stack_inst_f32_calib_i.hip',
    'stack_inst_f32_plain_i.hip',
    'stack_inst_u16_calib_i.hip',
    'stack_inst_u16_plain_i.hip',
    'stack_inst_f32_calib_d.hip',
    'stack_inst_f32_plain_d.hip',
    'stack_inst_u16_calib_d.hip',
    'stack_inst_u16_plain_d.hip',
    'stack_inst_f32_calib_m.hip',
    'stack_inst_f32_plain_m.hip',
    'stack_inst_u16_calib_m.hip',
    'stack_inst_u16_plain_m.hip',
    'stack_inst_f32_calib_c.hip',
    'stack_inst_f32_plain_c.hip',
    'stack_inst_u16_calib_c.hip',
    'stack_inst_u16_plain_c.hip',
    'stack_inst_f32_calib_l.hip',
    'stack_inst_f32_plain_l.hip',
    'stack_inst_u16_calib_l.hip',
    'stack_inst_u16_plain_l.hip',
    'stack_inst_f32_calib_b.hip',
    'stack_inst_f32_plain_b.hip',
    'stack_inst_u16_calib_b.hip',
    'stack_inst_u16_plain_b.hip',
    'stack_inst_f32_calib_k.hip',
    'stack_inst_f32_plain_k.hip',
    'stack_inst_u16_calib_k.hip',
    'stack_inst_u16_plain_k.hip',
    'stack_inst_f32_calib_a.hip',
    'stack_inst_f32_plain_a.hip',
    'stack_inst_u16_calib_a.hip',
    'stack_inst_u16_plain_a.hip']
HEADERS = ['common.h', 'stack_sort.h', 'stack_calibrate.h', 'stack_reduce.h', 'stack_kernels.h', os.path.join(ROOT, 'include', 'apgpu.h')]

# -ffp-contract=off: the reference's NumPy expressions round after every operation, so no FMA
# contraction anywhere; fused operations are written explicitly (fma()) where wanted.
HIPCC_FLAGS = (['-DAPGPU_DEVELOPMENT'] if os.environ.get('APGPU_DEVELOPMENT') else []) + ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off',
               '-fno-fast-math', '-Wall', '-Wno-unused-function', '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC]


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return 'hipcc'


def _newest(paths):
    return max(os.path.getmtime(p if os.path.isabs(p) else os.path.join(CSRC, p)) for p in paths)


# The stack kernels' translation units: machine LICM off - the complete lean kernel is one loop (plain launch and redo pass share
# its body) and hoisted loop-invariant values cost it 5-10 VGPRs, the difference between three and two wavefronts per SIMD.
STACK_TU_FLAGS = ['-mllvm', '-disable-machine-licm']


def _compile(src):
    obj = os.path.join(OBJ, src.replace('.hip', '.o'))
    srcp = os.path.join(CSRC, src)
    extra = ['stack_mad.h'] if src.startswith('stack_mad') else []
    if src.startswith('resample'):
        extra.append('resample_core.h')
    if src == 'stack_mad_wide.hip':
        extra.append('stack_mad.hip')                        # (the wide units include the narrow ones)
    if src == 'stack_mad_pairs_wide.hip':
        extra.append('stack_mad_pairs.hip')
    dep_time = max(os.path.getmtime(srcp), _newest(HEADERS + extra))
    if os.path.exists(obj) and os.path.getmtime(obj) >= dep_time:
        return obj, False
    cmd = [_hipcc()] + HIPCC_FLAGS + (STACK_TU_FLAGS if src.startswith(('stack_inst_', 'resample_stack')) else []) + ['-c', srcp, '-o', obj]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed for %s:\n%s' % (src, r.stdout[-4000:]))
    return obj, True


def build_library(force=False, verbose=False, jobs=None):
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    jobs = jobs or min(len(SOURCES), os.cpu_count() or 4)
    with cf.ThreadPoolExecutor(jobs) as ex:
        results = list(ex.map(_compile, SOURCES))
    objs = [o for o, _ in results]
    rebuilt = any(r for _, r in results)
    if rebuilt or not os.path.exists(LIB):
        cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s' % r.stdout[-4000:])
    if verbose:
        print('libapgpu.so %s' % ('rebuilt' if rebuilt else 'up to date'), LIB)
    return LIB


if __name__ == '__main__':
    build_library(force='--force' in sys.argv, verbose=True)
