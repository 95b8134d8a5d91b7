"""How much of a session's kernel time overlaps with a kernel of the OTHER utterance in flight?

    python tools/overlap_trace.py run [inflight=2] [utterances=12]     the workload (config-2 item, resident
                                                                       PCM, UtterancePipeline as in a session)
    python tools/overlap_trace.py parse <kernel_trace.csv>             the summary of a rocprofv3 kernel trace

On a GPU box:
    rocprofv3 --kernel-trace --output-format csv -d /tmp/ov -o t -- python tools/overlap_trace.py run 2 12
    python tools/overlap_trace.py parse $(find /tmp/ov -name '*kernel_trace.csv')
"""
import collections
import csv
import re
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def run(inflight, n):
    import numpy as np
    from pb_chime5_amd import ops, synthetic
    from pb_chime5_amd._capi import default_context
    ctx = default_context(0)
    params = ops.make_params(wpe=True, wpe_taps=10, wpe_delay=2, wpe_iterations=3, bss_iterations=20,
                             bss_iterations_post=1)
    ops._prepare_windows(ctx, params.stft_size, params.stft_shift)
    utt = synthetic.config2(seed=2)
    pcm = np.clip(np.round(utt.obs * 32768.0), -32768, 32767).astype(np.int16)
    cs = utt.ex['start_orig']['original']
    pipe = ops.UtterancePipeline(params, depth=inflight, first_ctx=ctx)
    import time
    for rep in range(2):                       # the second pass is the one to look at
        t0 = time.perf_counter()
        for i in range(n):
            if pipe.full():
                pipe.pop()
            pipe.enqueue(i, pcm, utt.activity_array, utt.target_index, cs, cs)
        while len(pipe):
            pipe.pop()
        dt = time.perf_counter() - t0
    print(f'inflight {inflight}: {1e3 * dt / n:.3f} ms per utterance')
    pipe.close()


def label(name):
    m = re.search(r'(\w+)_kernel', name)
    return m.group(1) if m else name[:30]


def parse(path):
    rows = []
    for r in csv.DictReader(open(path)):
        start = int(r.get('Start_Timestamp') or r.get('Start'))
        end = int(r.get('End_Timestamp') or r.get('End'))
        q = r.get('Stream_Id') or r.get('Queue_Id') or '0'
        rows.append((start, end, q, label(r['Kernel_Name'])))
    rows.sort()
    # the second half of the trace (the warm pass)
    rows = rows[len(rows) // 2:]
    t_begin, t_end = rows[0][0], max(r[1] for r in rows)
    # sweep: at every boundary, which kernels run
    events = []
    for i, (s, e, q, lab) in enumerate(rows):
        events.append((s, 1, i))
        events.append((e, 0, i))
    events.sort()
    active = set()
    last = events[0][0]
    busy = 0
    alone = collections.defaultdict(int)
    shared = collections.defaultdict(int)
    for t, kind, i in events:
        dt = t - last
        if dt > 0 and active:
            busy += dt
            queues = {rows[j][2] for j in active}
            for j in active:
                (shared if len(queues) > 1 else alone)[rows[j][3]] += dt
        last = t
        if kind:
            active.add(i)
        else:
            active.discard(i)
    total = sum(e - s for s, e, _, _ in rows)
    span = t_end - t_begin
    print(f'span {span / 1e6:.2f} ms, some kernel running {busy / 1e6:.2f} ms ({busy / span:.3f}), '
          f'sum of kernel durations {total / 1e6:.2f} ms ({total / span:.3f} x the span)')
    print('%-18s %10s %10s %8s' % ('kernel', 'alone ms', 'shared ms', 'shared'))
    for lab in sorted(set(alone) | set(shared), key=lambda l: -(alone[l] + shared[l])):
        a, s = alone[lab] / 1e6, shared[lab] / 1e6
        print('%-18s %10.2f %10.2f %8.2f' % (lab, a, s, s / max(a + s, 1e-9)))


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 2, int(sys.argv[3]) if len(sys.argv) > 3 else 12)
    else:
        parse(sys.argv[2])
