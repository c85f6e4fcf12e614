#!/usr/bin/env python3
"""The command line BAM -> BAM with the record path on the GPU (--record-path=gpu, the default), with the blocks inflated by the
host's threads instead (--bgzf-inflate=host), and with the records on the host's threads (--record-path=host): wall clock of the
whole process on 4 M and 16 M records, the tool's own stage lines, and the outputs' records compared.
usage: tools/cli_records_bench.py [--workdir DIR] [--unique 250000] [--small 16] [--large 64]"""
import argparse, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--workdir", default="/tmp/pvc3_records_bench")
	ap.add_argument("--unique", type=int, default=250_000)
	ap.add_argument("--small", type=int, default=16)
	ap.add_argument("--large", type=int, default=64)
	ap.add_argument("--skip-digest", action="store_true")
	args = ap.parse_args()
	import cli_e2e
	P = cli_e2e.prepare(args.workdir, scale=args.unique / 1_000_000.0, repeat=args.small)
	n_small = P["n"]
	n_large = cli_e2e.write_repeated(P, os.path.join(args.workdir, "in_large.bam"), args.large)

	def run(label, alignments, n, extra, out="out.bam", times=2):
		cmd = cli_e2e.project_command(P, alignments=alignments, output=out) + extra
		best, lines = None, []
		for _ in range(times):
			cli_e2e.fresh_output(cmd, args.workdir)
			t = time.perf_counter()
			p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=args.workdir)
			dt = time.perf_counter() - t
			if p.returncode:
				sys.stderr.write(p.stderr.decode()[-3000:])
				raise SystemExit("command failed: " + " ".join(cmd))
			if best is None or dt < best:
				best = dt
				lines = [l.strip() for l in p.stderr.decode().split("\n") if l.startswith("\tStages") or l.startswith("\tRecord path") or l.startswith("\tTimeline")]
		print("%-46s %8d records  %.3f s = %6.2f M records/s" % (label, n, best, n / best / 1e6), flush=True)
		for l in lines:
			print("      " + l, flush=True)
		return best

	run("warm-up (page cache, first GPU start-up)", "in.bam", n_small, [], times=1)
	t_empty = run("input without records", "empty.bam", 0, [], out="empty_out.bam")
	results = {}
	for label, extra in (("record path on the GPU", []), ("... blocks inflated by the host (--bgzf-inflate=host)", ["--bgzf-inflate=host"]), ("records on the host (--record-path=host)", ["--record-path=host"])):
		ts = run(label, "in.bam", n_small, extra, out="out_%d.bam" % len(results))
		tl = run(label, "in_large.bam", n_large, extra, out="out_large.bam")
		results[label] = (ts, tl)
		print("      marginal rate (large - small): %.2f M records/s; without the empty run's %.3f s: %.2f / %.2f M records/s"
			  % ((n_large - n_small) / max(1e-9, tl - ts) / 1e6, t_empty, n_small / max(1e-9, ts - t_empty) / 1e6, n_large / max(1e-9, tl - t_empty) / 1e6), flush=True)
	if not args.skip_digest:
		d = [cli_e2e.records_digest(os.path.join(args.workdir, "out_%d.bam" % k)) for k in range(3)]
		print("records of the three outputs: %s, %s" % (d[0][0], "identical" if d[0] == d[1] == d[2] else "DIFFERENT"), flush=True)
		if not (d[0] == d[1] == d[2]) or d[0][0] != n_small:
			raise SystemExit("the outputs differ")


if __name__ == "__main__":
	main()
