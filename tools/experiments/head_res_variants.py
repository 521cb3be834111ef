"""Compile-time variants of csrc/head_res.hip (built here, run on the GPU box), as warp_variants.py."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, 'tools', 'experiments', '_build')
VARIANTS = {'full': [], 'nodma': ['-DMTR_RES_DEBUG=1'], 'noepilogue': ['-DMTR_RES_DEBUG=4'],
            'nodma_noepilogue': ['-DMTR_RES_DEBUG=5']}


def build():
    sys.path.insert(0, ROOT)
    from metrabs_amd import build as b
    b.build_library(verbose=False)
    os.makedirs(OUT, exist_ok=True)
    others = [os.path.join(b.BUILD_DIR, f + '.o') for f in b.sources() if f != 'head_res.hip']
    for name, defs in VARIANTS.items():
        obj = os.path.join(OUT, f'res_{name}.o')
        subprocess.run([b._hipcc(), *b.FLAGS, *defs, '-c', os.path.join(b.CSRC, 'head_res.hip'), '-o', obj], check=True,
                       stderr=subprocess.DEVNULL)
        subprocess.run([b._hipcc(), '-shared', '-fPIC', f'--offload-arch={b.ARCH}', *others, obj, '-o',
                        os.path.join(OUT, f'libmtr_res_{name}.so')], check=True)
    print('built', len(VARIANTS))


if __name__ == '__main__':
    if sys.argv[1] == 'build':
        build()
    else:
        for name in VARIANTS:
            code = ("import sys, os; sys.path.insert(0, %r); from metrabs_amd import _lib; _lib.load(%r); "
                    "sys.argv = ['x', 'quick']; import runpy; runpy.run_path(%r, run_name='__main__')"
                    % (ROOT, os.path.join(OUT, f'libmtr_res_{name}.so'),
                       os.path.join(ROOT, 'tools', 'experiments', 'head16_res_probe.py')))
            print('==', name, flush=True)
            r = subprocess.run(['timeout', '120', sys.executable, '-c', code], capture_output=True, text=True)
            print(r.stdout[-1500:], r.stderr[-400:] if r.returncode else '', flush=True)
