"""Dev tool: instruction mix of the MFMA-densest basic block of each kernel in a hipcc -S listing."""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\w+):\s*; @", s, re.M)]
for i, (pos, name) in enumerate(starts):
    if pat not in name:
        continue
    end = starts[i + 1][0] if i + 1 < len(starts) else len(s)
    body = s[pos:end].split("s_endpgm")[0]
    blocks = re.split(r"\n(\.LBB\d+_\d+):", body)
    best = max(((blocks[j], len(re.findall(r"v_mfma", blocks[j + 1])), blocks[j + 1]) for j in range(1, len(blocks), 2)), key=lambda t: t[1])
    ins = [l.strip().split()[0] for l in best[2].split("\n") if l.strip() and not l.strip().startswith((";", "."))]
    print(name, best[0], "instructions", len(ins))
    print("   ", dict(Counter(ins).most_common(30)))
