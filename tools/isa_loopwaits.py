"""Dev tool: s_waitcnt / s_barrier / scratch instructions inside the steady-state K loop (the block with the LDS-DMA issues and a
back edge) of every gemm_bf16_kernel instance in a hipcc -S listing.  A compiler-inserted vmcnt(0) in there drains the LDS-DMA
pipeline on every K block.  usage: isa_loopwaits.py <file.s>"""
import re, sys
src = open(sys.argv[1]).read()
for m in re.finditer(r'^(_ZN\S*gemm_bf16_kernelI\S*):', src, re.M):
    name = m.group(1)
    body = src[m.end():]
    body = body[:body.index('s_endpgm')]
    blocks = re.split(r'\n(?=\.LBB\d+_\d+:)', body)
    for b in blocks:
        lab = b.split(':')[0].strip()
        nm, nd = b.count('v_mfma'), b.count('global_load_lds')
        if nm >= 32 and nd >= 6:      # the peeled first K block (counted wait) and the steady-state loop body
            waits = re.findall(r'(s_waitcnt [^\n;]*|s_barrier|scratch_\w+)', b)
            short = re.sub(r'_ZN12_GLOBAL__N_116gemm_bf16_kernelI|EEvNS_10GemmParamsEPKl', '', name)
            print(f"{short:22s} {lab:10s} mfma {nm} dma {nd}: " + ', '.join(w.strip() for w in waits))

# the LDS-DMA statements write M0 without restoring it: report anything else that touches it
other = [l.strip() for l in src.split('\n') if re.search(r'\bm0\b', l.split(';')[0]) and not re.match(r'\s*s_mov_b32 m0, s\d+', l)]
print(f"other M0 users in the listing: {len(other)}" + ("" if not other else "  e.g. " + other[0]))
