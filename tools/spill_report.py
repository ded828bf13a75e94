#!/usr/bin/env python3
"""Register / scratch report of every kernel of the product library: hipcc -Rpass-analysis=kernel-resource-usage over mageslam_amd/csrc/*.hip
(no GPU needed).  A kernel with a non-zero scratch size or vector-register spill count is listed first; exit code 1 when there is one
(tests/test_abi.py::test_no_kernel_spills_to_scratch pins that at zero).
    python tools/spill_report.py [--all]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


# kernels whose scratch is an indexed private array of the algorithm, not a spill: the tether edges' numerical differentiation
# (one thread per tether, central differences over a 12-entry state, BaseMultiEdge's way: DESIGN.md section 4.4)
PRIVATE_ARRAYS = ("k_tether_linearize",)


def report():
    from mageslam_amd import build as B
    rows = []
    for src in B.sources():
        p = subprocess.run([B.HIPCC, *B.flags_for(src), "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.devnull], capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError(p.stderr[-2000:])
        cur = None
        for line in p.stderr.splitlines():
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                cur = {"file": os.path.basename(src), "kernel": m.group(1)}
                rows.append(cur)
                continue
            m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill|Occupancy \[waves/SIMD\]|TotalSGPRs): (\d+)", line)
            if m and cur is not None:
                cur[m.group(1).split(" [")[0]] = int(m.group(2))
    return rows


def demangle(names):
    try:
        out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
        return [re.sub(r"\(mage::.*|\((?!anonymous).*", "", o.replace("mage::(anonymous namespace)::", "").replace("void ", "")) or o for o in out]
    except OSError:
        return names


def main():
    rows = report()
    names = demangle([r["kernel"] for r in rows])
    bad = 0
    print(f"{'kernel':58s} {'file':22s} VGPR  SGPR  occ  scratch  vspill sspill")
    for r, n in sorted(zip(rows, names), key=lambda x: (-(x[0].get("ScratchSize", 0) + x[0].get("VGPRs Spill", 0)), x[0]["file"], x[1])):
        spills = r.get("VGPRs Spill", 0) or (r.get("ScratchSize", 0) and not any(k in n for k in PRIVATE_ARRAYS))
        bad += 1 if spills else 0
        if spills or "--all" in sys.argv:
            print(f"{n[:58]:58s} {r['file']:22s} {r.get('VGPRs', 0):4d}  {r.get('TotalSGPRs', 0):4d}  {r.get('Occupancy', 0):3d}  {r.get('ScratchSize', 0):7d}  {r.get('VGPRs Spill', 0):6d} {r.get('SGPRs Spill', 0):6d}")
    print(f"{len(rows)} kernels, {bad} with spilled vector registers or unexplained scratch")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
