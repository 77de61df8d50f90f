"""Where the __amd_rocclr_copyBuffer dispatches of a bench.py run come from (58 per RCAN step in the round-4 profile).
Reads a rocprofv3 --kernel-trace CSV and counts the copy kernels between consecutive optimizer launches (adam_pack_kernel = one per training
step), so that set-up (plan construction: tables and zeroed buffers uploaded once), the timed steps and bench.py's `as_called` leg (host tensors
in, image back to the host: the reference caller's form) can be told apart.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/copytrace -o p -- python3 bench.py --model rcan --steps 20 --warmup 5 --probe-steps 1 --no-cpu-baseline
    python3 tests/tools/copy_attrib.py gpurun_out/copytrace/p_kernel_trace.csv 20 5
"""
import csv
import sys


def main(path, steps, warmup):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    per, cnt, t_copy = [], 0, 0
    for r in rows:
        if 'copyBuffer' in r['Kernel_Name']:
            cnt += 1
            t_copy += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        elif 'adam_pack_kernel' in r['Kernel_Name']:
            per.append(cnt)
            cnt = 0
    total = sum(per) + cnt
    print('%d copyBuffer dispatches in the run (%.1f us of GPU time); %d optimizer launches = training steps' % (total, t_copy / 1e3, len(per)))
    if not per:
        return
    print('  before the first step ends (plan construction: packed tables, zeroed buffers, the first batch): %d' % per[0])
    timed = per[1:warmup + steps]                       # steps 2 .. W + K of the contract's region (step 1 carries the set-up)
    if timed:
        print('  per step over the contract\'s W + K region (steps 2 .. %d): min %d, max %d, mean %.2f  <- what the timed step contains'
              % (warmup + steps, min(timed), max(timed), sum(timed) / len(timed)))
    rest = per[warmup + steps:]
    if rest:
        print('  per step behind it (settled probe, kernel probe, `as_called` leg): ' + ' '.join(str(v) for v in rest))
    print('  behind the last step: %d' % cnt)


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 20, int(sys.argv[3]) if len(sys.argv) > 3 else 5)
