"""Build the C part of the oracle (test infrastructure) with gcc:
oracle/libgrl_oracle.so.  Called by __graft_entry__.build(); building the checker
is not using it."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, 'libgrl_oracle.so')


def build(force=False):
    srcs = [os.path.join(HERE, 'ref_c', f) for f in ('gemm_chain.c', 'jpeg_baseline.c')]
    if (not force and os.path.isfile(OUT)
            and all(os.path.getmtime(OUT) >= os.path.getmtime(s) for s in srcs)):
        return OUT
    subprocess.check_call(['gcc', '-O2', '-mfma', '-ffp-contract=off', '-fopenmp', '-shared',
                           '-fPIC'] + srcs + ['-o', OUT, '-lm'])
    return OUT


if __name__ == '__main__':
    print(build(force=True))
