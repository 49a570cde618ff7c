"""Per-kernel register / LDS / code-size table of the product library, from the compiler's own resource report (no GPU needed):

    python tools/kernel_resources.py > profiles/r06_kernel_resources.txt

Columns: VGPRs + AGPRs one wave allocates (gfx950: 512 per SIMD lane, shared by every resident wave of the SIMD), waves per SIMD one workgroup
brings (threads / 256), the share of a SIMD's register file ONE workgroup takes, static LDS, code bytes.  A CU hosts two workgroups only if
their register shares AND their LDS both fit -- profiles/r06_notes.md section 2 reads the step's collisions off this table."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'vslnet_amd', 'csrc')
sys.path.insert(0, ROOT)
from vslnet_amd.build import FLAGS, FILE_FLAGS, SOURCES, _hipcc      # noqa: E402

THREADS = {'k_convblock_fwd2': 512, 'k_convblock_fwd': 512, 'k_convblock_bwd': 512, 'k_attn_block_fwd<1': 1024, 'k_attn_block_fwd<2': 512, 'k_attn_bwd_fused': 1024,
           'k_attn_bwd_long': 1024, 'k_wgrad4': 512, 'k_wgrad3': 256, 'k_vproj_fwd3': 512, 'k_embed_fwd': 512, 'k_embed_bwd': 512, 'k_query_fwd': 256, 'k_query_bwd': 256}


def demangle(names):
    r = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True)
    return r.stdout.split('\n')


def main():
    rows = []
    for src in SOURCES:
        if src == 'api.hip':
            continue
        cmd = [_hipcc()] + [f for f in FLAGS if f != '-fPIC'] + FILE_FLAGS.get(src, []) + ['--cuda-device-only', '-S', os.path.join(CSRC, src), '-o', '/tmp/_kr.s']
        subprocess.run(cmd, check=True, capture_output=True)
        txt = open('/tmp/_kr.s').read()
        for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', txt, re.S):
            name, body = m.group(1), m.group(2)
            lds = int(re.search(r'\.amdhsa_group_segment_fixed_size (\d+)', body).group(1))
            after = txt[m.end():m.end() + 4000]
            info = txt[txt.find('.size\t' + name):]
            code = int(re.search(r'; codeLenInByte = (\d+)', info).group(1))
            v = int(re.search(r'; NumVgprs: (\d+)', info).group(1))
            a = int(re.search(r'; NumAgprs: (\d+)', info).group(1))
            scr = int(re.search(r'; ScratchSize: (\d+)', info).group(1))
            rows.append((src, name, v, a, lds, code, scr))
    dn = demangle([r[1] for r in rows])
    print('# compiler resource report of vslnet_amd/csrc (hipcc --offload-arch=gfx950, the build\'s flags); tools/kernel_resources.py')
    print('# regs = VGPRs + AGPRs per lane of one wave (allocated in blocks of 8; 512 per SIMD lane); w/SIMD = waves one workgroup puts on a SIMD;')
    print('# share = w/SIMD x ceil8(regs) / 512 of the register file ONE workgroup takes; LDS = static bytes (dynamic LDS: see the launchers)')
    print('%-92s %5s %5s %6s %6s %8s %7s %7s' % ('kernel', 'VGPR', 'AGPR', 'w/SIMD', 'share', 'LDS', 'code B', 'scratch'))
    for (src, name, v, a, lds, code, scr), d in zip(rows, dn):
        short = re.sub(r'\(.*', '', d).replace('void vsl::', '').replace('vsl::', '')
        thr = 256
        for k, t in THREADS.items():
            if short.startswith(k):
                thr = t
        wps = thr // 256
        regs = (v + a + 7) // 8 * 8
        print('%-92s %5d %5d %6d %5.0f%% %8d %7d %7d' % (short[:92], v, a, wps, 100.0 * wps * regs / 512, lds, code, scr))


if __name__ == '__main__':
    main()
