#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel, read from the code objects of the built object files (the ELF notes: what the
hardware is given -- rocprofv3's `vgpr` field is half the ELF's count on gfx950).
usage: python profiles/kernel_resources.py [build/obj] [name filter]"""
import glob, os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernels_of(obj):
    out = []
    with tempfile.TemporaryDirectory() as td:
        co, fb = os.path.join(td, "dev.co"), os.path.join(td, "dev.fatbin")
        r = subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fb, obj], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(fb):
            return out
        r = subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fb, "--output=" + co],
                           capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
            return out
        txt = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        for blk in re.split(r"\n  - (?=\.agpr_count)", txt)[1:]:
            cur = {}
            for m in re.finditer(r"^    \.(\w+):\s+(\S+)\s*$", blk, re.M):
                cur[m.group(1)] = m.group(2)
            if "name" in cur:
                out.append(cur)
    return out


def main():
    d = sys.argv[1] if len(sys.argv) > 1 else "build/obj"
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    seen = set()
    print("%-70s %5s %6s %6s %6s %8s %8s" % ("kernel", "VGPR", "vspill", "SGPR", "sspill", "LDS B", "scratch"))
    for obj in sorted(glob.glob(os.path.join(d, "*.o"))):
        for k in kernels_of(obj):
            name = subprocess.run(["c++filt", k.get("name", "?")], capture_output=True, text=True).stdout.strip() or k.get("name", "?")
            name = re.sub(r"^void augx::dev::|^void ", "", name)
            name = re.sub(r"\(.*$", "", name)
            if flt and flt not in name:
                continue
            if name in seen:
                continue
            seen.add(name)
            print("%-70s %5s %6s %6s %6s %8s %8s" % (name[:70], k.get("vgpr_count"), k.get("vgpr_spill_count"), k.get("sgpr_count"), k.get("sgpr_spill_count"),
                                                      k.get("group_segment_fixed_size"), k.get("private_segment_fixed_size")))


if __name__ == "__main__":
    main()
