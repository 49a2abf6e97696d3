"""Build helpers: compile pase_amd/csrc/*.hip for gfx950 into pase_amd/libpase_hip.so (in-tree, so
the .so travels to the GPU box with the gpurun snapshot), and -- for the CPU kernel tests only --
the same sources against the SIMT emulator into tests/hipemu/libpase_emu.so.
"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pase_amd", "csrc")
INCLUDE = os.path.join(ROOT, "include")
HIP_SO = os.path.join(ROOT, "pase_amd", "libpase_hip.so")
# the same library with the staging waves' loads left to the compiler's own s_waitcnt bookkeeping (-DPASE_X6C_AUTOWAIT):
# test infrastructure only -- tests/test_conv_x6c.py compares it bit for bit with the shipped hand-counted waits on the GPU
AUTOWAIT_SO = os.path.join(ROOT, "tests", "libpase_hip_autowait.so")
RESOURCES = os.path.join(ROOT, "pase_amd", "libpase_hip.resources.json")
EMU_DIR = os.path.join(ROOT, "tests", "hipemu")
EMU_SO = os.path.join(EMU_DIR, "libpase_emu.so")


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    return sorted(hdrs)


def _up_to_date(out, digest):
    stamp = out + ".sha256"
    return os.path.exists(out) and os.path.exists(stamp) and open(stamp).read().strip() == digest


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("build failed: " + " ".join(cmd))
    return r.stdout


def _hip_flags():
    # -fno-slp-vectorize: the SLP vectoriser pairs adjacent fp32 adds / muls / fmas into v_pk_*_f32, which beside a
    # stream of MFMAs costs 20+ cycles per instruction instead of ~4 (MI355X_MICROARCH.md, "price of one filler beside
    # MFMAs"); the staging code of every GEMM kernel here (affine + PReLU + bf16 residuals) is exactly that pattern
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize", "-I", INCLUDE, "-I", CSRC,
            "-Wno-unused-result"]


def hip_digest():
    # path-independent (the gpurun snapshot lives under a scratch root): file contents + the flags without the -I paths
    return _digest(_sources() + _deps(), " ".join(f for f in _hip_flags() if f not in (INCLUDE, CSRC)))


def _kernel_resources(remarks):
    """{mangled kernel name: {VGPRs, ScratchSize, VGPRs Spill, SGPRs Spill, LDS Size, ...}} from hipcc's
    -Rpass-analysis=kernel-resource-usage remarks."""
    import re
    res, cur = {}, None
    for line in remarks.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = res.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass-analysis", line)
        if m and cur is not None:
            try:
                cur[m.group(1).strip()] = int(m.group(2))
            except ValueError:
                cur[m.group(1).strip()] = m.group(2)
    return res


def check_x6c_resources(res):
    """The staging waves of conv_x6c_kernel's register-staged instantiations (template argument ZP = false) issue their loads
    as inline asm whose destination VGPRs the compiler believes defined at issue, waited for by hand-counted s_waitcnt and
    tied to their first use by x6c_claim (conv_x6c.hip, x6c_gload).  A scratch spill or a register copy of those VGPRs before
    the wait would read stale data silently, and the kernel sits at 249 of 256 VGPRs: a toolchain or flag change that makes
    it spill must fail the BUILD, not a training run.  Raises on any private-segment use or spilled VGPR there."""
    bad = []
    seen = 0
    for name, r in res.items():
        if "conv_x6c_kernelILi" not in name:
            continue
        seen += 1
        # conv_x6c_kernel<NPOS, KGS, TM, ZP, NARROW>: ...ILi<NPOS>ELi<KGS>ELb<TM>ELb<ZP>ELb<NARROW>EE...
        import re
        m = re.search(r"conv_x6c_kernelILi(\d+)ELi(\d+)ELb([01])ELb([01])ELb([01])E", name)
        zp = bool(m and m.group(4) == "1")
        if zp:
            continue          # pre-split operands are staged by LDS DMA: no asm-loaded registers
        if r.get("ScratchSize", 0) != 0 or r.get("VGPRs Spill", 0) != 0:
            bad.append((name, r.get("ScratchSize"), r.get("VGPRs Spill")))
    if seen == 0:
        raise RuntimeError("build: no conv_x6c_kernel instantiation in the resource remarks (flag or name changed?)")
    if bad:
        raise RuntimeError("build: conv_x6c_kernel instantiations with hand-counted staging waits use scratch / spill VGPRs "
                           "(stale-register hazard, see check_x6c_resources): %r" % (bad,))


def check_x6c_staging_isa(asm_text):
    """ISA lint of conv_x6c.hip's hidden staging loads (x6c_gload: inline asm, destination considered defined at issue,
    waited for by hand-counted s_waitcnt, handed to the conversion by x6c_claim).  Loads and claims carry their register
    set -- and the claims their operand registers -- in the asm text.  Per kernel:
      * every load destination belongs to exactly ONE set and every set has the same number of destinations (a load that
        lands in a temporary shows up as a destination shared between sets or as extra destinations);
      * the registers a set's claims hand to the conversion are exactly the registers its loads were issued into (a value
        that was copied between issue and claim -- vector assembly, live-range split, merged sibling branches -- is read
        from a register the load never wrote: STALE on the hardware, invisible to the emulator).
    Round 5 found both failure classes in the first forms of a streamed variant of the loop (since removed); this is the
    build-time tripwire."""
    import re

    def regs_of(tok):
        m = re.match(r"v\[(\d+):(\d+)\]$", tok)
        if m:
            return list(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.match(r"v(\d+)$", tok)
        return [int(m.group(1))] if m else []
    kern, cur = {}, None
    for line in asm_text.splitlines():
        m = re.match(r"\s*\.type\s+(\S*conv_x6c_kernel\S*),@function", line)
        if m:
            cur = kern.setdefault(m.group(1), dict(load={}, claim={}))
            continue
        if cur is None:
            continue
        m = re.match(r"\s*global_load_dword(?:x2|x4)?\s+(v\[\d+:\d+\]|v\d+),.*; staging set (\d+)", line)
        if m:
            for r in regs_of(m.group(1)):
                cur["load"].setdefault(r, set()).add(int(m.group(2)))
            continue
        m = re.match(r"\s*; claim staging set (\d+) regs (.*)$", line)
        if m:
            for tok in m.group(2).split():
                for r in regs_of(tok):
                    cur["claim"].setdefault(int(m.group(1)), set()).add(r)
    # third rule: no COMPILER-inserted `s_waitcnt vmcnt` inside a hidden load sequence (within six lines of a hidden load and not
    # itself inline asm).  The compiler knows nothing of the hidden loads: such a wait protects a register of some pending
    # VISIBLE load -- e.g. never-consumed weight-fragment prefetches of the compute waves, whose exit path the compiler merged
    # with the role branch, or per-column parameters loaded in setup_item -- and in doing so drains the hidden pipeline on
    # every tick (round 5: 14 ... 80 % on the strided / 1x1 launches).  The kernels close those books with visible waits.
    lines = asm_text.splitlines()
    sync_waits = {}
    cur_name = None
    hidden_at = []
    for i, line in enumerate(lines):
        m = re.match(r"\s*\.type\s+(\S*conv_x6c_kernel\S*),@function", line)
        if m:
            cur_name = m.group(1)
            hidden_at = []
            continue
        if cur_name is None:
            continue
        if "; staging set" in line and "global_load" in line:
            hidden_at.append(i)
        elif re.match(r"\s*s_waitcnt vmcnt", line) and "#ASMSTART" not in lines[i - 1]:
            sync_waits.setdefault(cur_name, []).append(i)
    hidden_by_kernel = {}
    cur_name = None
    for i, line in enumerate(lines):
        m = re.match(r"\s*\.type\s+(\S*conv_x6c_kernel\S*),@function", line)
        if m:
            cur_name = m.group(1)
        elif cur_name is not None and "; staging set" in line and "global_load" in line:
            hidden_by_kernel.setdefault(cur_name, []).append(i)
    bad = []
    for name, waits in sync_waits.items():
        hid = hidden_by_kernel.get(name, [])
        import bisect
        n = 0
        for w in waits:
            j = bisect.bisect_left(hid, w)
            if any(0 <= jj < len(hid) and abs(hid[jj] - w) <= 6 for jj in (j - 1, j)):
                n += 1
        if n:
            bad.append((name, "%d compiler-inserted s_waitcnt vmcnt inside hidden load sequences" % n))
    checked = 0
    for name, k in kern.items():
        if not k["load"]:
            continue
        checked += 1
        shared = sorted(r for r, sets in k["load"].items() if len(sets) > 1)
        per_set = {}
        for r, sets in k["load"].items():
            for st in sets:
                per_set.setdefault(st, set()).add(r)
        if shared or len(set(len(v) for v in per_set.values())) > 1:
            bad.append((name, "load destinations shared between sets %r, per set %r"
                        % (shared[:8], {st: len(v) for st, v in per_set.items()})))
        for st, regs in per_set.items():
            cl = k["claim"].get(st, set())
            if cl != regs:
                bad.append((name, "set %d: loaded into %r but claimed from %r"
                            % (st, sorted(regs - cl)[:12], sorted(cl - regs)[:12])))
    if checked == 0:
        raise RuntimeError("build: no hidden staging load found in the conv_x6c assembly (asm text changed?)")
    if bad:
        raise RuntimeError("build: conv_x6c staging loads are not claimed from the registers they were issued into "
                           "(stale-register hazard, see check_x6c_staging_isa): %r" % (bad,))
    return checked


def build_hip(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU).

    Order matters: the shipped library is linked and stamped only AFTER the resource check and the staging-load ISA lint of
    conv_x6c.hip have passed -- a build that fails either leaves no stamped .so behind for _lib._check_fresh() to accept.  The
    lint runs on EVERY build (PASE_BUILD_NO_AUTOWAIT=1 only skips the second, compiler-waited compile that the bit-for-bit
    GPU test needs)."""
    import json
    srcs = _sources()
    flags = _hip_flags()
    digest = hip_digest()
    # PASE_BUILD_NO_AUTOWAIT=1 (development iterations): skip the second compile of conv_x6c.hip; the GPU test that needs
    # tests/libpase_hip_autowait.so then refuses a stale one by its digest stamp
    want_aw = os.environ.get("PASE_BUILD_NO_AUTOWAIT", "0") != "1"
    if not force and _up_to_date(HIP_SO, digest) and (not want_aw or _up_to_date(AUTOWAIT_SO, digest)):
        return HIP_SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    need_main = force or not _up_to_date(HIP_SO, digest)
    objs = [s[:-4] + ".o" for s in srcs]
    cflags = [f for f in flags if f != "-shared"]
    x6c = os.path.join(CSRC, "conv_x6c.hip")
    aw_obj = os.path.join(CSRC, "conv_x6c.autowait.o")
    lint_s = os.path.join(CSRC, "conv_x6c.lint.s")
    running = []

    def spawn(cmd):
        pr = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        running.append(pr)
        return pr

    def reap(pr, what):
        out, _ = pr.communicate()
        running.remove(pr)
        if pr.returncode != 0:
            sys.stderr.write(out)
            raise RuntimeError("hipcc failed on " + what)
        return out

    def invalidate(so):
        for f in (so, so + ".sha256"):
            if os.path.exists(f):
                os.remove(f)

    try:
        procs = []
        if need_main:
            invalidate(HIP_SO)            # nothing stale may survive a failed build
            for s, o in zip(srcs, objs):
                extra = ["-Rpass-analysis=kernel-resource-usage"] if s == x6c else []
                procs.append(spawn([hipcc, "-c", s, "-o", o] + cflags + extra))
        # (in parallel: the device assembly of conv_x6c.hip for the staging-load lint -- every build -- and conv_x6c.hip once
        #  more with the compiler's own waits, for the bit-for-bit GPU test)
        lint = spawn([hipcc, "-S", "--cuda-device-only", x6c, "-o", lint_s] + [f for f in cflags if f != "-fPIC"]) \
            if need_main else None
        aw = None
        if want_aw:
            invalidate(AUTOWAIT_SO)
            aw = spawn([hipcc, "-c", x6c, "-o", aw_obj, "-DPASE_X6C_AUTOWAIT"] + cflags)
        for pr, s in zip(procs, srcs):
            out = reap(pr, s)
            if s == x6c:
                res = _kernel_resources(out)
                check_x6c_resources(res)
                with open(RESOURCES, "w") as f:
                    json.dump({k: v for k, v in sorted(res.items()) if "conv_x6c_kernel" in k}, f, indent=1)
            elif verbose and out:
                print(out)
        if lint is not None:
            reap(lint, "conv_x6c.hip (-S, staging-load lint)")
            with open(lint_s) as f:
                check_x6c_staging_isa(f.read())
            os.remove(lint_s)
        if need_main:
            _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", HIP_SO] + objs)
            with open(HIP_SO + ".sha256", "w") as f:
                f.write(digest)
        if aw is not None:
            reap(aw, "conv_x6c.hip (-DPASE_X6C_AUTOWAIT)")
            missing = [o for o in objs if not os.path.exists(o)]
            if missing:
                raise RuntimeError("build: object files of the shipped library are missing (%r); rebuild with --force" % missing[:3])
            _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", AUTOWAIT_SO] +
                 [aw_obj if o.endswith("conv_x6c.o") else o for o in objs])
            with open(AUTOWAIT_SO + ".sha256", "w") as f:
                f.write(digest)
    except BaseException:
        # a failed check must not leave compilers running behind the raised error (nor a half-written library)
        for pr in list(running):
            pr.kill()
            pr.communicate()
        raise
    return HIP_SO


def build_emu(force=False):
    """g++ build of the same kernel sources against tests/hipemu (CPU kernel tests only)."""
    srcs = _sources()
    emu_srcs = [os.path.join(EMU_DIR, "hipemu.cpp")]
    flags = ["-O2", "-std=c++17", "-fPIC", "-shared", "-DPASE_HIPEMU", "-I", INCLUDE, "-I", CSRC, "-I", EMU_DIR,
             "-pthread", "-Wno-unused-result", "-fno-strict-aliasing"]
    digest = _digest(srcs + _deps() + emu_srcs + [os.path.join(EMU_DIR, "hipemu.h")],
                     " ".join(f for f in flags if f not in (INCLUDE, CSRC, EMU_DIR)))
    if not force and _up_to_date(EMU_SO, digest):
        return EMU_SO
    objs = []
    procs = []
    for s in srcs + emu_srcs:
        o = os.path.join(EMU_DIR, os.path.basename(s) + ".emu.o")
        objs.append(o)
        cmd = ["g++", "-x", "c++", "-c", s, "-o", o] + [f for f in flags if f != "-shared"]
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for pr, s in zip(procs, srcs + emu_srcs):
        out, _ = pr.communicate()
        if pr.returncode != 0:
            sys.stderr.write(out)
            raise RuntimeError("g++ (emu) failed on " + s)
    _run(["g++", "-shared", "-fPIC", "-pthread", "-o", EMU_SO] + objs)
    with open(EMU_SO + ".sha256", "w") as f:
        f.write(digest)
    return EMU_SO


if __name__ == "__main__":
    which = sys.argv[1:] or ["hip"]
    if "hip" in which:
        print(build_hip(force="--force" in which, verbose=True))
    if "emu" in which:
        print(build_emu(force="--force" in which))
