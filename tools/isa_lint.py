#!/usr/bin/env python
"""ISA lint of the built library: extracts the gfx950 code objects from libgrl_hip.so (clang offload bundles in
.hip_fatbin), disassembles them with llvm-objdump and reports, per kernel, the instruction count, MFMA count, scratch
(spill) instructions and how many of its `s_waitcnt vmcnt(N)` are N == 0.

Why: hipcc counts vmcnt per basic block -- a branch inside an epilogue or a staging loop silently turns every wait into
vmcnt(0) (round 5: 458 of the 463 waits of gemm_f32_kernel<128,128>; -1.5 % of the headline) -- and a register-pressure
regression shows up as scratch traffic long before it shows up in a test.  tests/test_boundary_cpu.py runs `lint()` on
every build.

  python tools/isa_lint.py [libgrl_hip.so] [name-substring ...]
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def code_objects(path, arch='gfx950'):
    blob = open(path, 'rb').read()
    out, pos = [], 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            break
        n = struct.unpack_from('<Q', blob, pos + len(MAGIC))[0]
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from('<QQQ', blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if arch in triple and size > 0:
                out.append(blob[pos + off:pos + off + size])
        pos += len(MAGIC)
    return out


def kernels(path):
    """{demangled-ish kernel symbol: dict(instr, mfma, scratch, vm0, vm)}"""
    res = {}
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix='.co') as f:
            f.write(co)
            f.flush()
            txt = subprocess.run([OBJDUMP, '-d', '--demangle', '--no-show-raw-insn', f.name], stdout=subprocess.PIPE,
                                 stderr=subprocess.DEVNULL, check=True).stdout.decode(errors='replace')
        cur, body = None, []

        def close():
            # "between the MFMAs": from the first to the last v_mfma of the kernel -- its k loops (and, in kernels with
            # MFMAs in the epilogue, whatever lies between them)
            if cur is None:
                return
            mf = [i for i, x in enumerate(body) if x.startswith('v_mfma')]
            if mf:
                seg = body[mf[0]:mf[-1] + 1]
                cur['scratch_mm'] += sum(1 for x in seg if x.startswith('scratch_'))
                cur['vm0_mm'] += sum(1 for x in seg if x.startswith('s_waitcnt') and 'vmcnt(0)' in x)
        for line in txt.splitlines():
            m = re.match(r'^[0-9a-f]+ <(.+)>:$', line)
            if m:
                close()
                cur = res.setdefault(m.group(1), dict(instr=0, mfma=0, scratch=0, vm0=0, vm=0, scratch_mm=0, vm0_mm=0))
                body = []
                continue
            if cur is None or not line.startswith('\t') and not line.startswith(' '):
                continue
            ins = line.strip()
            if not ins or ins.startswith('//'):
                continue
            body.append(ins)
            cur['instr'] += 1
            if ins.startswith('v_mfma'):
                cur['mfma'] += 1
            elif ins.startswith('scratch_'):
                cur['scratch'] += 1
            elif ins.startswith('s_waitcnt') and 'vmcnt(' in ins:
                cur['vm'] += 1
                if 'vmcnt(0)' in ins:
                    cur['vm0'] += 1
        close()
    return res


# The rule (round 5; holds for all 236 kernels of the library but the two below): between the first and the last MFMA of a
# kernel -- its k loops -- there is no scratch instruction (no register spill inside an MFMA loop) and at most two
# `s_waitcnt vmcnt(0)` (the loops wait with counted vmcnt; a stage barrier's own drain is allowed).
MIN_MFMA = 8
ALLOW = {   # known offenders: (max scratch, max vmcnt(0)) between the MFMAs
    'gemm_bf16_256_kernel<false, true, false, true, false, 2>': (8, 9),    # BN-reduce epilogue with the mask recomputed from z: MFMAs of the statistics sit behind it
    'bneck_tail_kernel<128, 512, 256, 64, 2, 2, 16, 0>': (4, 4),           # layer 2 -> 3 fused tail (one launch per step): 256 VGPRs, four spills per chunk
}


def lint(path):
    ks = kernels(path)
    bad = []
    for k, v in ks.items():
        if v['mfma'] < MIN_MFMA:
            continue
        ms, mv = 0, 2
        for sub, lim in ALLOW.items():
            if sub in k:
                ms, mv = lim
        if v['scratch_mm'] > ms:
            bad.append('%s: %d scratch instructions between its MFMAs (register spills in a k loop)' % (k[:110], v['scratch_mm']))
        if v['vm0_mm'] > mv:
            bad.append('%s: %d x s_waitcnt vmcnt(0) between its MFMAs (a branch or a dependent instruction behind a load?)' % (k[:110], v['vm0_mm']))
    return ks, bad


if __name__ == '__main__':
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith('.so') else os.path.join(here, 'grl_amd', 'libgrl_hip.so')
    subs = [a for a in sys.argv[1:] if not a.endswith('.so')]
    ks, bad = lint(lib)
    print('%-100s %7s %6s %7s %9s %9s %9s' % ('kernel', 'instr', 'mfma', 'scratch', 'vmcnt0/all', 'scr@mfma', 'vm0@mfma'))
    for k in sorted(ks, key=lambda k: -ks[k]['instr']):
        if subs and not any(s in k for s in subs):
            continue
        v = ks[k]
        if not subs and v['instr'] < 200:
            continue
        print('%-100s %7d %6d %7d %5d/%-4d %8d %9d' % (k[:100], v['instr'], v['mfma'], v['scratch'], v['vm0'], v['vm'], v['scratch_mm'], v['vm0_mm']))
    for b in bad:
        print('LINT:', b)
    sys.exit(1 if bad else 0)
