"""Developer builds of libmetrabs_hip.so that differ from the product in ONE source's defines:
    python tools/experiments/variant_lib.py <name> <source.hip> -DX=1 [-DY=2 ...]
-> tools/experiments/_build/libmtr_<name>.so = that source recompiled with the defines + the product build's objects of
every other source (seconds, not a whole-library compile).  Load it with metrabs_amd._lib.load(path) / MTR_PROBE_LIB."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, 'tools', 'experiments', '_build')


def build(name, source, defs):
    os.makedirs(OUT, exist_ok=True)
    sys.path.insert(0, ROOT)
    from metrabs_amd import build as product
    product.build_library(verbose=False)
    others = [os.path.join(product.BUILD_DIR, f + '.o') for f in product.sources() if f != source]
    obj = os.path.join(OUT, f'{name}.o')
    subprocess.run(['hipcc', *product.FLAGS, *product.EXTRA_FLAGS.get(source, []), *defs, '-c',
                    os.path.join(product.CSRC, source), '-o', obj], check=True, stderr=subprocess.DEVNULL)
    lib = os.path.join(OUT, f'libmtr_{name}.so')
    subprocess.run(['hipcc', '-shared', '-fPIC', f'--offload-arch={product.ARCH}', obj, *others, '-o', lib], check=True)
    os.remove(obj)
    return lib


if __name__ == '__main__':
    print(build(sys.argv[1], sys.argv[2], sys.argv[3:]))
