#!/usr/bin/env python3
"""One run of the command line (record path on the GPU unless --arg says otherwise) on the 16 M-record file with everything it reports, and the wall clock around it.
usage: tools/cli_once.py [--workdir DIR] [--repeat N] [KEY=VALUE ...]"""
import argparse, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--workdir", default="/tmp/pvc3_records_bench")
	ap.add_argument("--repeat", type=int, default=3)
	ap.add_argument("--copies", type=int, default=64)
	ap.add_argument("--keep-output", action="store_true", help="overwrite the previous run's output instead of writing a new file")
	ap.add_argument("--arg", action="append", default=[], help="an argument more for the command line, e.g. --arg=--record-path=host")
	ap.add_argument("env", nargs="*")
	args = ap.parse_args()
	import cli_e2e
	P = cli_e2e.prepare(args.workdir, scale=0.25, repeat=1)
	n = cli_e2e.write_repeated(P, os.path.join(args.workdir, "in_large.bam"), args.copies)
	cmd = cli_e2e.project_command(P, alignments="in_large.bam", output="out_large.bam") + args.arg
	env = dict(os.environ, **dict(kv.split("=", 1) for kv in args.env))
	subprocess.run(cmd, cwd=args.workdir, capture_output=True, env=env)
	for _ in range(args.repeat):
		if not args.keep_output: cli_e2e.fresh_output(cmd, args.workdir)
		t_launch = time.time()
		t = time.perf_counter()
		p = subprocess.run(cmd, cwd=args.workdir, capture_output=True, env=env)
		dt = time.perf_counter() - t
		print("%.3f s = %.2f M records/s; launched at %.3f, back at %.3f" % (dt, n / dt / 1e6, t_launch, time.time()))
		print(p.stderr.decode())
		sys.stdout.flush()


if __name__ == "__main__":
	main()
