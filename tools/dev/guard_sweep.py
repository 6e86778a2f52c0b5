"""Runs the GPU test suite one process per test FILE under the diagnostic allocator modes of the library (csrc/devmem.cpp:
HFNET_GUARD_ALLOC=1 / 2 = page-guarded allocations, HFNET_GUARD_FILL=xx = poisoned fresh memory); a file whose process dies
(a device fault aborts the host process) is re-run one process per TEST so that every faulting test is named.

    HFNET_GUARD_ALLOC=1 python tools/dev/guard_sweep.py [pytest -k expression]

Prints one line per file / test and a summary; exit code 1 if anything failed or died."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(args, timeout):
    try:
        r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + args, cwd=ROOT,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
        return r.returncode, r.stdout
    except subprocess.TimeoutExpired as e:
        return -9, (e.stdout or b"").decode(errors="replace") if isinstance(e.stdout, bytes) else (e.stdout or "")


def main():
    kexpr = sys.argv[1] if len(sys.argv) > 1 else None
    mode = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("HFNET_GUARD"))
    print(f"guard sweep [{mode or 'no guard mode set'}]", flush=True)
    rc, out = run(["--collect-only"] + (["-k", kexpr] if kexpr else []), 300)
    ids = [l.strip() for l in out.splitlines() if "::" in l and not l.startswith(" ")]
    files = sorted(set(i.split("::")[0] for i in ids))
    bad = []
    for f in files:
        rc, out = run([f] + (["-k", kexpr] if kexpr else []), 1500)
        tail = out.strip().splitlines()[-1] if out.strip() else ""
        print(f"{f}: rc {rc}  {tail}", flush=True)
        if rc in (0, 5):
            continue
        if rc == 1:                                      # an ordinary test failure: pytest names it
            bad.append((f, rc, out[-3000:]))
            print(out[-3000:], flush=True)
            continue
        for t in [i for i in ids if i.split("::")[0] == f]:          # the process died: one process per test
            rc2, out2 = run([t], 900)
            if rc2 not in (0, 5):
                fault = [l for l in out2.splitlines() if "Memory access fault" in l or "Aborted" in l or "HSA_STATUS" in l]
                print(f"    {t}: rc {rc2}  {fault[:2]}", flush=True)
                bad.append((t, rc2, out2[-1500:]))
    print(f"guard sweep [{mode}]: {len(files)} files, {len(bad)} failing / dying entries")
    for b in bad:
        print("BAD", b[0], b[1])
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
