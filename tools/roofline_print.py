import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['step_kernel_ms'])
for k in d['roofline']['kernels']: print(f"{k['kernel']:<46} n={k['launches_per_step']:<5} {k['avg_us']:>8.1f} us {k['ms_per_step']:.3f} ms  {k['tflops']:>6.1f} TF")
