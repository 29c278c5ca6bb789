"""A/B of the plan's LDS slot numbering (EGS_SLOT_BANKS=1 bank-aware, 0 first-use) on the
1-lane tile kernel: C4 fp32 x 1024 ensembles and C3 fp64 x 24 piles, isotropic-body variant
on and off.  Each case in a fresh process (the switch is read when the plan is built)."""
import json, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for wl, batch in (("c4", 1024), ("c3", 24)):
    for iso in ("0", None):
        for rep in range(2):
            for banks in ("1", "0"):
                env = dict(os.environ, EGS_SLOT_BANKS=banks)
                if iso is not None:
                    env["EGS_ISO"] = iso
                out = subprocess.run([sys.executable, os.path.join(R, "bench.py"), "--workload", wl, "--batch", str(batch),
                                      "--steps", "20", "--warmup", "3", "--cpu-seconds", "0", "--no-single"],
                                     env=env, capture_output=True, text=True, timeout=200).stdout
                d = json.loads(out.strip().splitlines()[-1])
                print(f"{wl} x{batch} iso={'default' if iso is None else iso} banks={banks}: "
                      f"kernel {d['roofline']['kernel_ms']:.4f} ms, step {d['ms_per_step']:.4f} ms", flush=True)
