"""Register / spill report of the kernels inside the built libalgames_hip.so (build check, no GPU needed).

The library is a HIP fat binary: every translation unit contributes one clang offload bundle to the `.hip_fatbin` section
(compressed with --offload-compress since round 5: unpacked here with clang-offload-bundler).
This module extracts the gfx950 code objects and reads the AMDGPU kernel metadata (`llvm-readelf --notes`): VGPR / SGPR counts,
scratch bytes and the spill counts of every kernel.  tests/test_abi.py uses it to keep the solver kernels spill-free."""
import os
import re
import struct
import subprocess
import tempfile

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


BUNDLER = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"
CMAGIC = b"CCOB"                # compressed offload bundle (--offload-compress): magic, u16 version, u16 method, then the sizes


def _compressed_bundles(blob, arch):
    """Device images of the compressed bundles in `blob` (one per translation unit), unpacked by clang-offload-bundler."""
    pos = 0
    while True:
        pos = blob.find(CMAGIC, pos)
        if pos < 0:
            return
        version, method = struct.unpack_from("<HH", blob, pos + 4)
        if version < 2 or version > 3 or method > 1:     # not a header (the four bytes can occur in data)
            pos += 4
            continue
        total = struct.unpack_from("<Q" if version == 3 else "<I", blob, pos + 8)[0]
        if total <= 24 or pos + total > len(blob):
            pos += 4
            continue
        with tempfile.TemporaryDirectory() as d:
            src, dst = os.path.join(d, "b.ccob"), os.path.join(d, "b.co")
            open(src, "wb").write(blob[pos:pos + total])
            r = subprocess.run([BUNDLER, "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--" + arch, "--input=" + src, "--output=" + dst],
                               capture_output=True, text=True)
            if r.returncode == 0 and os.path.exists(dst) and os.path.getsize(dst) > 0:
                yield open(dst, "rb").read()
        pos += total


def code_objects(lib_path, arch="gfx950"):
    """Yields the device ELF images (bytes) for `arch` contained in the fat binary (plain or compressed bundles)."""
    blob = open(lib_path, "rb").read()
    yield from _compressed_bundles(blob, arch)
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        n = struct.unpack_from("<Q", blob, pos + len(MAGIC))[0]
        cur = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, cur)
            triple = blob[cur + 24:cur + 24 + tlen].decode()
            cur += 24 + tlen
            if arch in triple and size > 0:
                yield blob[pos + off:pos + off + size]
        pos = cur


def kernel_resources(lib_path):
    """{demangled kernel name: dict(vgpr, sgpr, scratch, vgpr_spill, sgpr_spill, lds)} over all code objects."""
    out = {}
    for img in code_objects(lib_path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(img)
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for blk in txt.split("- .agpr_count:")[1:]:
            def g(key):
                m = re.search(r"\." + key + r":\s*(\S+)", blk)
                return m.group(1) if m else None
            name = g("name")
            if not name:
                continue
            out[name] = dict(vgpr=int(g("vgpr_count")), sgpr=int(g("sgpr_count")), scratch=int(g("private_segment_fixed_size")),
                             vgpr_spill=int(g("vgpr_spill_count")), sgpr_spill=int(g("sgpr_spill_count")),
                             lds=int(g("group_segment_fixed_size")))
    names = list(out)
    if names:
        dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.strip().split("\n")
        # (Cfg's sixth parameter -- line-search staging in LDS, default 1 -- is dropped from the names when it has its default, so that the names
        # of rounds 1-5 stay what tests and profiles know: Cfg<model, p, d, ext, waves>)
        out = {re.sub(r"(Cfg<\d+, \d+, \d+, \d+, \d+), 1>", r"\1>", re.sub(r"^void |alg::|\(.*$", "", d)): v for d, v in zip(dem, out.values())}
    return out


def scratch_load_counts(lib_path, only=None):
    """{demangled kernel name: number of scratch_load* instructions in its ISA} (llvm-objdump -d over the gfx950 code objects).
    A kernel whose metadata reports a private segment but whose ISA never loads from it only parks a by-reference argument
    object there (stores without loads): no register spill is involved."""
    objdump = os.path.join(os.path.dirname(READELF), "llvm-objdump")
    out = {}
    for img in code_objects(lib_path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(img)
            f.flush()
            txt = subprocess.run([objdump, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
        cur = None
        for line in txt.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                cur = m.group(1)
                out.setdefault(cur, 0)
            elif cur is not None and "scratch_load" in line:
                out[cur] += 1
    names = [n for n in out if n.startswith("_Z")]
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.strip().split("\n") if names else []
    # (Cfg's sixth parameter -- line-search staging in LDS, default 1 -- is dropped from the names when it has its default, so that the names of
    # rounds 1-5 stay what tests and profiles know: Cfg<model, p, d, ext, waves>)
    res = {re.sub(r"(Cfg<\d+, \d+, \d+, \d+, \d+), 1>", r"\1>", re.sub(r"^void |alg::|\(.*$", "", d)): out[n] for d, n in zip(dem, names)}
    return res if only is None else {k: v for k, v in res.items() if k in only}


if __name__ == "__main__":
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    res = kernel_resources(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "lib", "libalgames_hip.so"))
    pat = sys.argv[2] if len(sys.argv) > 2 else "k_newton_solve|k_ibr|k_mpc_loop"
    for k in sorted(res):
        if re.search(pat, k):
            v = res[k]
            print("%-52s vgpr %3d sgpr %3d scratch %4d vgpr_spill %3d sgpr_spill %3d lds %5d" % (k, v["vgpr"], v["sgpr"], v["scratch"], v["vgpr_spill"], v["sgpr_spill"], v["lds"]))
