"""Instruction mix of the K loop of each kernel in a hipcc -S listing (spills inside it are a red flag).

    python tools/kloop_stats.py file.s [name-substring]
"""
import re
import sys


def main(path, filt=""):
    text = open(path).read()
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", text, flags=re.S | re.M):
        name, body = m.group(1), m.group(2)
        if filt not in name:
            continue
        L = body.split("\n")
        bars = [i for i, l in enumerate(L) if "s_barrier" in l]
        if len(bars) < 2:
            continue
        labels = {}
        for i, l in enumerate(L):
            mm = re.match(r"^(\.LBB\d+_\d+):", l)
            if mm:
                labels[mm.group(1)] = i
        b = bars[1]
        start = end = None
        for i in range(b, len(L)):
            mm = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", L[i])
            if mm and labels.get(mm.group(1), 10 ** 9) < b:
                start, end = labels[mm.group(1)], i
                break
        if start is None:
            continue
        seg = L[start:end + 1]
        cnt = lambda pat: sum(bool(re.search(pat, l)) for l in seg)   # noqa: E731
        valu, salu = cnt(r"^\s+v_(?!mfma)"), cnt(r"^\s+s_")
        print(f"{name[:60]:60s} loop {start}-{end}: mfma={cnt('v_mfma')} scratch={cnt('scratch_')} "
              f"ds_read={cnt('ds_read')} glds={cnt('global_load_lds')} valu={valu} salu={salu} "
              f"total_scratch={sum('scratch_' in l for l in L)}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
