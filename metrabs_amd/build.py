"""Builds libmetrabs_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m metrabs_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so lands next to the sources
(metrabs_amd/csrc/libmetrabs_hip.so): it is git-ignored but travels with gpurun snapshots.
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB_PATH = os.path.join(CSRC, 'libmetrabs_hip.so')
BUILD_DIR = os.path.join(CSRC, 'build')
ARCH = 'gfx950'
FLAGS = ['-O3', '-std=c++17', '-fPIC', f'--offload-arch={ARCH}', '-Wall', '-Wno-unused-function',
         '-Wno-inline-asm']  # (the LDS-DMA asm declares its M0 clobber; clang warns that M0 is a reserved register)
# per-source extras.  head_rt.hip: MFMA accumulators in VGPRs (gfx950's register file is unified):
# its f32 chains are carried into f64 on the VALU every stage, and from AGPRs every element costs a
# v_accvgpr_read first.
EXTRA_FLAGS = {'head_rt.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form=1']}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))


def _hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found; metrabs_amd needs ROCm to build its HIP kernels')
    return exe


def _digest(path):
    h = hashlib.sha256()
    h.update(' '.join(FLAGS + EXTRA_FLAGS.get(os.path.basename(path), [])).encode())
    for dep in [path] + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith('.h')] \
            + [os.path.join(os.path.dirname(os.path.dirname(CSRC)), 'include', 'metrabs_hip.h')]:
        with open(dep, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def _compile(src, force):
    path = os.path.join(CSRC, src)
    obj = os.path.join(BUILD_DIR, src + '.o')
    stamp = obj + '.sha256'
    digest = _digest(path)
    if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == digest:
        return obj, False
    cmd = [_hipcc(), *FLAGS, *EXTRA_FLAGS.get(src, []), '-c', path, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'hipcc failed for {src}:\n{r.stdout}\n{r.stderr}')
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    with open(stamp, 'w') as f:
        f.write(digest)
    return obj, True


def build_library(force=False, verbose=True):
    os.makedirs(BUILD_DIR, exist_ok=True)
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in results]
    rebuilt = any(r for _, r in results)
    if rebuilt or force or not os.path.exists(LIB_PATH):
        cmd = [_hipcc(), '-shared', '-fPIC', f'--offload-arch={ARCH}', *objs, '-o', LIB_PATH]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
        if verbose:
            print(f'built {LIB_PATH} from {len(srcs)} sources')
    elif verbose:
        print(f'{LIB_PATH} is up to date')
    return LIB_PATH


def kernel_resources(src):
    """Register / scratch / LDS footprint of every kernel of one source file, read from the code
    object metadata the compiler emits (hipcc -S --cuda-device-only, same flags as the build).
    -> list of dicts: name (mangled), vgpr_count, agpr_count, vgpr_spill_count, sgpr_spill_count,
    private_segment_fixed_size (scratch bytes per lane), group_segment_fixed_size (static LDS)."""
    import re
    import tempfile
    path = os.path.join(CSRC, src)
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, src + '.s')
        cmd = [_hipcc(), *[f for f in FLAGS if f != '-fPIC'], *EXTRA_FLAGS.get(src, []), '-S',
               '--cuda-device-only', path, '-o', asm]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc -S failed for {src}:\n{r.stderr}')
        text = open(asm).read()
    loops = mfma_loop_lane_traffic(text)
    out = []
    meta = text[text.index('amdhsa.kernels:'):] if 'amdhsa.kernels:' in text else ''
    for blk in re.split(r'\n  - ', meta)[1:]:
        if '.vgpr_count:' not in blk:
            continue  # (amdhsa.version's list items)
        get = lambda k: int(re.search(rf'\.{k}:\s+(\d+)', blk).group(1))
        name = re.search(r'\.name:\s+(\S+)', blk).group(1)
        out.append(dict(name=name, mfma_loops=loops.get(name, []),
                        **{k: get(k) for k in ('vgpr_count', 'agpr_count', 'vgpr_spill_count',
                                               'sgpr_spill_count', 'private_segment_fixed_size',
                                               'group_segment_fixed_size')}))
    return out


def mfma_loop_lane_traffic(asm_text):
    """Per kernel of an assembly listing: its INNERMOST loops that contain MFMAs (a backward branch whose
    range holds no other backward branch -- the K loops), each as dict(mfma, lane_moves, barriers, dma):
    lane_moves = v_readlane + v_writelane instructions inside the loop, i.e. SGPRs the register allocator
    parked in VGPR lanes and restores on every pass.  The compiler spills dozens of uniform values of the
    head kernels (kernel arguments, addresses of the decode epilogue); this is how a test pins that none
    of that traffic sits in a K loop, where it would take VALU slots beside the MFMAs."""
    import re
    out = {}
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n', asm_text, flags=re.M):
        name = m.group(1)
        end = asm_text.find('.Lfunc_end', m.end())
        if end < 0:
            continue
        body = asm_text[m.end():end].split('\n')
        labels = {}
        for i, line in enumerate(body):
            lm = re.match(r'^(\.LBB\d+_\d+):', line)
            if lm:
                labels[lm.group(1)] = i
        spans = []
        for i, line in enumerate(body):
            bm = re.search(r's_c?branch\S*\s+(\.LBB\d+_\d+)', line)
            if bm and labels.get(bm.group(1), 1 << 30) <= i:
                spans.append((labels[bm.group(1)], i))
        inner = [sp for sp in spans if not any(o != sp and sp[0] <= o[0] and o[1] <= sp[1] for o in spans)]
        loops = []
        for a, b in inner:
            seg = body[a:b + 1]
            n = lambda key: sum(key in x for x in seg)
            if n('v_mfma'):
                loops.append(dict(mfma=n('v_mfma'), lane_moves=n('v_readlane') + n('v_writelane'),
                                  barriers=n('s_barrier'), dma=n('global_load_lds')))
        out[name] = loops
    return out


if __name__ == '__main__':
    build_library(force='--force' in sys.argv)
