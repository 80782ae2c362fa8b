#!/usr/bin/env python3
"""What can this GPU's HBM take in stores? Context for track mode, whose kernels write 0.7 of their traffic.
Device-side timing (events) of: a fill, a copy, eight fills at once on eight streams, and a scattered fill that writes
2 KiB runs at random places (the store pattern of one k_tracks20s wave: 64 lanes x 32 bytes per track and iteration)."""
import json
import torch

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

def main():
    dev = torch.device("cuda:0")
    n = 1 << 29  # doubles: 4 GiB
    x = torch.empty(n, dtype=torch.float64, device=dev)
    y = torch.empty(n, dtype=torch.float64, device=dev)
    out = {}
    ms = timed(lambda: x.fill_(1.0)); out["fill_4GiB_TBps"] = 8 * n / ms / 1e9
    ms = timed(lambda: y.copy_(x)); out["copy_4GiB_TBps_read_plus_write"] = 16 * n / ms / 1e9
    parts = x.view(8, -1)
    streams = [torch.cuda.Stream() for _ in range(8)]
    def eight():
        for k, s in enumerate(streams):
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s): parts[k].fill_(2.0)
        for s in streams: torch.cuda.current_stream().wait_stream(s)
    ms = timed(eight); out["eight_fills_on_eight_streams_TBps"] = 8 * n / ms / 1e9
    # scattered 2 KiB runs: index_fill of rows of 256 doubles in a random order
    rows = x.view(-1, 256)
    perm = torch.randperm(rows.shape[0], device=dev)
    src = torch.ones(256, dtype=torch.float64, device=dev)
    def scattered(): rows[perm] = src
    ms = timed(scattered, reps=3); out["scattered_2KiB_runs_TBps_stores_only"] = 8 * n / ms / 1e9
    # read-modify-free mix like a track kernel: read 1 byte per 64 written is negligible; a read + write mix at 1:3
    z = torch.empty(n // 3, dtype=torch.float64, device=dev)
    def mix():
        x.fill_(3.0); z.sum()
    ms = timed(mix); out["fill_then_read_a_third_TBps"] = (8 * n + 8 * (n // 3)) / ms / 1e9
    print(json.dumps(out))

if __name__ == "__main__":
    main()
