#!/usr/bin/env python3
"""The command line's record path on the GPU with the GPU's share of a piece's blocks (PVC3_RECORDS_GPU_BLOCKS; the host's threads inflate
the rest) and the inflater limited to fewer blocks at a time (PVC3_RECORDS_INFLATE_IN_FLIGHT): room in the CUs' LDS for the other kernels against inflater throughput.
usage: tools/cli_inflight_sweep.py [--workdir DIR]"""
import argparse, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--workdir", default="/tmp/pvc3_records_bench")
	args = ap.parse_args()
	import cli_e2e
	P = cli_e2e.prepare(args.workdir, scale=0.25, repeat=1)
	n = cli_e2e.write_repeated(P, os.path.join(args.workdir, "in_large.bam"), 64)
	cmd = cli_e2e.project_command(P, alignments="in_large.bam", output="out_large.bam")
	subprocess.run(cmd, cwd=args.workdir, capture_output=True)
	for env in ({}, {"PVC3_RECORDS_GPU_BLOCKS": "2048"}, {"PVC3_RECORDS_GPU_BLOCKS": "1536"}, {"PVC3_RECORDS_GPU_BLOCKS": "1024"}, {"PVC3_RECORDS_GPU_BLOCKS": "1024", "PVC3_RECORDS_INFLATE_IN_FLIGHT": "1024"},
				{"PVC3_RECORDS_GPU_BLOCKS": "2048", "PVC3_RECORDS_INFLATE_IN_FLIGHT": "2048"}, {"PVC3_RECORDS_INFLATE_IN_FLIGHT": "2048"}):
		best, line = None, ""
		for _ in range(2):
			cli_e2e.fresh_output(cmd, args.workdir)
			t = time.perf_counter()
			p = subprocess.run(cmd, cwd=args.workdir, capture_output=True, env=dict(os.environ, **env))
			dt = time.perf_counter() - t
			if p.returncode:
				print(env, "FAILED", p.stderr.decode()[-500:]); break
			if best is None or dt < best:
				best = dt
				line = [l.strip() for l in p.stderr.decode().split("\n") if l.startswith("\tStages")][0]
		print("%-80s %.3f s = %.2f M records/s\n      %s" % (env, best, n / best / 1e6, line), flush=True)


if __name__ == "__main__":
	main()
