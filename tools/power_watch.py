#!/usr/bin/env python3
"""`python tools/power_watch.py [--every 0.3] -- <command ...>`: runs the command as a child and samples `rocm-smi` beside it
(socket power, sclk / mclk / fclk, temperature, throttle status where reported): is a bandwidth-bound kernel running against the
power cap or at reduced clocks?  The watcher itself never touches the GPU.  One JSON summary line on stderr at the end."""
import json
import subprocess
import sys
import threading
import time


def sample():
    try:
        out = subprocess.run(["rocm-smi", "-P", "-c", "-t", "-u", "--json"], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out)
        card = d.get("card0") or next(iter(d.values()))
        return {k: v for k, v in card.items()}
    except Exception as e:      # noqa: BLE001
        return {"error": str(e)}


def main():
    argv = sys.argv[1:]
    every = 0.3
    if argv and argv[0] == "--every":
        every = float(argv[1])
        argv = argv[2:]
    if argv and argv[0] == "--":
        argv = argv[1:]
    samples = []
    stop = threading.Event()

    def loop():
        while not stop.is_set():
            s = sample()
            s["t"] = time.time()
            samples.append(s)
            stop.wait(every)

    th = threading.Thread(target=loop, daemon=True)
    th.start()
    rc = subprocess.call(argv)
    stop.set()
    th.join(timeout=15)
    keys = {}
    for s in samples:
        for k, v in s.items():
            if k in ("t", "error"):
                continue
            try:
                num = float(str(v).strip("()").replace("Mhz", "").replace("MHz", ""))
            except ValueError:
                continue
            keys.setdefault(k, []).append(num)
    summary = {k: {"min": min(v), "max": max(v), "mean": sum(v) / len(v), "n": len(v)} for k, v in keys.items()}
    print(json.dumps({"power_watch": summary, "samples": len(samples), "first_raw": samples[0] if samples else None,
                      "errors": sum(1 for s in samples if "error" in s)}), file=sys.stderr, flush=True)
    sys.exit(rc)


if __name__ == "__main__":
    main()
