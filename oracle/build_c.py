"""TEST INFRASTRUCTURE: builds the plain-C oracle (oracle/mtr_oracle.c) with gcc into
oracle/_build/libmtr_oracle.so (git-ignored; travels with gpurun snapshots)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'mtr_oracle.c')
OUT_DIR = os.path.join(HERE, '_build')
LIB = os.path.join(OUT_DIR, 'libmtr_oracle.so')


def build(verbose=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    # -ffp-contract=off: keep the reference's unfused float op order
    cmd = ['gcc', '-O2', '-std=c99', '-fPIC', '-shared', '-ffp-contract=off', SRC, '-lm', '-o', LIB]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'gcc failed:\n{r.stderr}')
    if verbose:
        print('built', LIB)
    return LIB


if __name__ == '__main__':
    build(verbose=True)
