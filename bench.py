#!/usr/bin/env python3
"""bench.py — throughput of the project_alignments hot path on MI355X (BASELINE.json metric).

A step = one pass of the whole hot path (projection walk, mate projection, indel-run realignment, packing) over one
device-resident batch of synthetic records (default: config C3 of BASELINE.json, 20 M 2x150 bp pairs = 40 M records per GPU,
REF + 8 founders, 1 Gbp in 8 contigs).  Inputs are resident in HBM when the timed region starts.

Multi-GPU: one process per GPU, every rank owns its own slice of the read stream (weak scaling), the index is replicated,
no data-path collective.  `python bench.py --gpus N` starts the N ranks itself (children are fresh processes, started before
anything in this process touches the GPU); under `python -m torch.distributed.run` the ranks are already there and this file
is one of them.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0		# MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling

METRIC = {
	"C2": "projected+realigned alignments/sec, 150 bp single-end",
	"C3": "projected+realigned alignments/sec, 2x150 bp",
	"C3G": "projected+realigned alignments/sec, 2x150 bp, gap-dense MSA",
	"C4": "projected+realigned alignments/sec, 2x150 bp",
	"C5": "projected+realigned alignments/sec, 2x250 bp, 5 % long-indel reads",
}


def usable_cpus():
	"""The CPUs this process can really use: the affinity mask, cut down by a cgroup CPU quota (the GPU boxes show every core of
	their host but grant a share of them — `cores` of the CPU baseline must be that share, not the host's core count)."""
	n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
	try:
		quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
		if quota != "max" and int(period) > 0:
			n = min(n, max(1, -(-int(quota) // int(period))))
	except Exception:
		pass
	return max(1, n)


def log(msg):
	if int(os.environ.get("RANK", "0")) == 0:
		print("[bench] " + msg, file=sys.stderr, flush=True)


def parse_args(argv=None):
	ap = argparse.ArgumentParser()
	ap.add_argument("--gpus", type=int, default=1)
	ap.add_argument("--steps", type=int, default=10)
	ap.add_argument("--warmup", type=int, default=2)
	ap.add_argument("--config", default="C3", help="BASELINE.json workload: C2, C3 (metric config), C3G (C3 on a gap-dense MSA), C4 (its per-GPU shard: 50 M records), C5")
	ap.add_argument("--scale", type=float, default=1.0, help="shrink genome and read count together (smoke runs only)")
	ap.add_argument("--cpu-baseline-seconds", type=float, default=15.0, help="target CPU time of the oracle sample; 0 disables")
	ap.add_argument("--score-mode", type=int, default=0)
	ap.add_argument("--option", action="append", default=[], metavar="KEY=VALUE", help="pvc3_context_set_option (tuning experiments), e.g. realign_waves=5120")
	ap.add_argument("--cli-records", type=int, default=4_000_000, help="records of the end-to-end command line leg (BAM file -> panvc3_project_alignments -> BAM file, SURVEY.md §8d protocol iii); 0 disables")
	ap.add_argument("--skip-host-buffers", action="store_true", help="leave out the PCIe-inclusive legs (their smaller launches would enter rocprofv3's per-kernel averages)")
	ap.add_argument("--sorted", action="store_true", help="sort the records by (source entry, position) like a coordinate-sorted BAM (secondary data point; the default is the unsorted worst case)")
	ap.add_argument("--skip-sorted-leg", action="store_true", help="leave out the second measurement of the same records in coordinate order (the order of a sorted BAM, which is what the workflows feed project_alignments)")
	ap.add_argument("--other-configs", default="C5,C3G,C3:position_windows=0", help="further configurations measured at full size after the main one (a few steps each, never part of `value`): other_configs in the JSON line; CONFIG or CONFIG:option=value,... (pvc3_context_set_option; C3:position_windows=0 is the index without position windows, i.e. k_project_fast_probes); empty disables")
	ap.add_argument("--other-steps", type=int, default=5)
	ap.add_argument("--setup-only", action="store_true", help="run the set-up of every rank (shared genome, this rank's reads, the host side of the index), print its seconds and peak resident set, and stop before anything touches a GPU (ranks then meet over gloo): rehearsal of the N-rank launch on a box without N GPUs")
	ap.add_argument("--no-build", action="store_true", help="never compile: exit non-zero if a native library is stale (use under rocprofv3, after `python3 -m panvc3_amd.build`)")
	return ap.parse_args(argv)


def ensure_built(no_build, local_rank):
	"""All native pieces, before torch or any HIP call exists in this process (a compiler must never be started from a process
	that has initialised the GPU, least of all under rocprofv3).  Builds are keyed by content, so on the GPU box, where the
	libraries travel with the tree, this compiles nothing.  Only local rank 0 builds; the others wait for fresh keys."""
	from panvc3_amd import build as pvc3_build
	from oracle import oracle as orc
	from tests.synth import synth

	def stale():
		return [name for name, fn in (("libpvc3_hip.so", pvc3_build.is_stale), ("libpvc3_oracle.so", orc.is_stale), ("libpvc3_synth.so", synth.is_stale)) if fn()]
	s = stale()
	if not s:
		return
	if no_build:
		raise SystemExit("bench.py --no-build: stale native libraries %s; run `python3 -m panvc3_amd.build` (and the oracle / synth builds) first" % ", ".join(s))
	if 0 == local_rank:
		pvc3_build.build()
		orc.build()
		synth.build()
		return
	t0 = time.time()
	while stale():
		if time.time() - t0 > 900:
			raise SystemExit("bench.py: rank %d gave up waiting for local rank 0 to build" % local_rank)
		time.sleep(1.0)


def launch_children(args):
	"""`python bench.py --gpus N` without a launcher: this process becomes the parent of N rank processes and never touches
	the GPU itself (no torch import, no HIP call)."""
	from panvc3_amd.shard import launch_ranks
	ensure_built(args.no_build, 0)
	argv = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--no-build"]
	codes = launch_ranks(argv, args.gpus)
	bad = [(r, c) for r, c in enumerate(codes) if c != 0]
	if bad:
		print("[bench] ranks failed: %s" % ", ".join("rank %d exit %s" % rc for rc in bad), file=sys.stderr, flush=True)
		return 1
	return 0


def cli_end_to_end(n_records):
	"""Wall clock of bin/panvc3_project_alignments on a synthetic BAM of config C2's reads (REF + 2 founders, 150 bp single-end),
	BAM in, BAM out, the whole process each time: start-up, index file, FASTA, GPU contexts, the records, the output file."""
	import tempfile
	sys.path.insert(0, os.path.join(ROOT, "tools"))
	import cli_e2e
	here = os.getcwd()
	work = tempfile.mkdtemp(prefix="pvc3_bench_cli_")
	try:
		unique = min(n_records, 250_000)
		P = cli_e2e.prepare(work, scale=unique / 1_000_000.0, repeat=max(1, n_records // unique))
		cmd = cli_e2e.project_command(P)
		cli_e2e.timed(cmd, cwd=work)		# page cache, first GPU start-up
		runs = [cli_e2e.timed(cmd, cwd=work) for _ in range(3)]
		dt, stages = min(runs, key=lambda r: r[0])
		t_empty, _ = cli_e2e.timed(cli_e2e.project_command(P, alignments="empty.bam", output="empty_out.bam"), cwd=work)
		# The same on four times the records: what a run costs whatever its size (start-up, and the exit of a process that has
		# used the GPU) is half a second, half of the small run.
		n_large = cli_e2e.write_repeated(P, os.path.join(work, "in_large.bam"), 4 * max(1, n_records // unique))
		cmd_large = cli_e2e.project_command(P, alignments="in_large.bam")
		dt_large = min(cli_e2e.timed(cmd_large, cwd=work)[0] for _ in range(2))
		# the records on the host's threads instead (--record-path=host: round 3's path, the comparator of the GPU's record path)
		dt_host, stages_host = min((cli_e2e.timed(cmd + ["--record-path=host"], cwd=work) for _ in range(2)), key=lambda r: r[0])
		dt_host_large = min(cli_e2e.timed(cmd_large + ["--record-path=host"], cwd=work)[0] for _ in range(2))
		# sixteen times the records: what a run costs whatever its size no longer decides
		n_huge = cli_e2e.write_repeated(P, os.path.join(work, "in_huge.bam"), 16 * max(1, n_records // unique))
		os.remove(os.path.join(work, "in_large.bam"))
		dt_huge = cli_e2e.timed(cli_e2e.project_command(P, alignments="in_huge.bam"), cwd=work)[0]
		return {
			"value": P["n"] / dt, "unit": "records/s", "records": P["n"], "seconds": dt, "seconds_without_records": t_empty,
			"cores": usable_cpus(), "stages": stages,
			"large": {"records": n_large, "seconds": dt_large, "value": n_large / dt_large},
			"huge": {"records": n_huge, "seconds": dt_huge, "value": n_huge / dt_huge},
			"marginal": {"value": (n_large - P["n"]) / max(1e-9, dt_large - dt), "unit": "records/s", "what": "(records of large - records of the small run) / (their seconds' difference)"},
			"record_path_host": {"seconds": dt_host, "value": P["n"] / dt_host, "stages": stages_host, "large": {"seconds": dt_host_large, "value": n_large / dt_host_large},
								 "marginal": (n_large - P["n"]) / max(1e-9, dt_host_large - dt_host)},
			"what": "bin/panvc3_project_alignments, BAM file in, BAM file out, wall clock of the whole process, best of 3 (large: best of 2, huge: one run): the records are "
					"inflated, found, projected, written anew and compressed on the GPU (--record-path=gpu, the default); "
					"record_path_host: the same with the records on the host's threads; seconds_without_records: the same command on an input without records (index file, FASTA, GPU start-up)",
		}
	finally:
		os.chdir(here)
		import shutil
		shutil.rmtree(work, ignore_errors=True)


def traffic_for(config, scale):
	"""HBM bytes per k_project_fast launch from the committed PMC passes, but only if they were taken with the kernels that
	are running now (profiles/pmc_traffic.json records the build key); otherwise null — never a stale constant."""
	from panvc3_amd import build as pvc3_build
	prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
	try:
		pj = json.load(open(prof))
		ent = pj.get("%s@%g" % (config, scale))
		if ent and ent.get("kernel_key") == pvc3_build.kernel_key():
			return ent["k_project_fast_hbm_bytes_per_launch"], ent
	except Exception:
		pass
	return None, None


def measure_other_config(config, args, local_rank, threads, cores, options=()):
	"""One more BASELINE.json configuration at full size on this GPU: ms per step, kernel times, the whole step against the HBM
	roofline, and a sample of the results compared with the oracle.  Everything it makes is released again."""
	import numpy as np
	import torch
	import panvc3_amd
	from tests.synth import synth
	from oracle import oracle as orc
	gp, rp, desc = synth.config(config, scale=args.scale)
	t0 = time.time()
	genome = synth.Genome(gp, n_threads=min(threads, gp.n_contigs))
	reads = synth.Reads(genome, rp, n_threads=threads)
	batch = reads.batch()
	n = int(reads.n)
	t_gen = time.time() - t0
	index = panvc3_amd.Index()
	genome.fill_index(index)
	ctx = panvc3_amd.Context(local_rank)
	ctx.set_option("window_read_length", int(rp.read_len))
	for k, v in options:
		ctx.set_option(k, int(v))
	ctx.upload_index(index)
	ctx.set_timing(True)
	db = panvc3_amd.DeviceBatch(ctx, batch)
	params = panvc3_amd.api.make_params(0, -1, args.score_mode)
	try:
		for _ in range(2):
			db.project(params)
			stats = db.finish()
		keys = ("project", "mid", "general", "realign", "pack")
		ms = dict.fromkeys(keys, 0.0)
		steps = max(1, args.other_steps)
		torch.cuda.synchronize()
		tb = time.perf_counter()
		for _ in range(steps):
			db.project(params)
			st = db.finish()
			for k in ms:
				ms[k] += st["ms_" + k]
		torch.cuda.synchronize()
		dt = time.perf_counter() - tb
		ms_step = dt / steps * 1e3
		out = {
			"workload": desc, "records": n, "steps": steps, "ms_per_step": ms_step, "value": n * steps / dt, "unit": "alignments/s",
			"kernel_ms": {k: v / steps for k, v in ms.items()},
			"whole_step": {"algorithmic_bytes": stats["algorithmic_bytes"], "GBps": stats["algorithmic_bytes"] / (ms_step * 1e-3) / 1e9, "frac": stats["algorithmic_bytes"] / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBPS},
			"batch": {"fast_path_records": stats["n_fast_path"], "mid_path_records": stats["n_mid_path"], "realigned_records": stats["n_realigned_records"], "indel_runs": stats["n_runs"]},
			"setup_seconds": t_gen,
		}
		# parity: the first records of the same batch through the oracle (a few seconds of CPU)
		o = orc.Oracle()
		genome.fill_index(o)
		sample = int(min(n, 200_000))
		want = o.batch_project(synth.slice_batch(batch, 0, sample), 0, -1, args.score_mode, n_threads=cores)
		got = db.download()
		n_c = int(want["out_cigar_off"][sample])
		same = (
			np.array_equal(got["dst_pos"][:sample], want["dst_pos"]) and np.array_equal(got["mate_dst_pos"][:sample], want["mate_dst_pos"])
			and np.array_equal(got["out_cigar_off"][:sample + 1], want["out_cigar_off"]) and np.array_equal(got["out_cigar"][:n_c], want["out_cigar"])
			and np.array_equal(got["status"][:sample], want["status"])
			and np.array_equal(got["run_off"][:sample + 1], want["run_off"]) and np.array_equal(got["runs"][:int(want["run_off"][sample])], want["runs"])
		)
		out["parity_sample"] = {"records": sample, "bit_identical": bool(same)}
		if not same:
			raise SystemExit("parity failure against the oracle on the %s sample" % config)
		return out
	finally:
		db.close()
		ctx.close()


def main():
	args = parse_args()
	if args.gpus < 1:
		raise SystemExit("--gpus must be positive")
	if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
		sys.exit(launch_children(args))

	rank = int(os.environ.get("RANK", "0"))
	world = int(os.environ.get("WORLD_SIZE", "1"))
	local_rank = int(os.environ.get("LOCAL_RANK", "0"))
	if world != args.gpus:
		raise SystemExit("bench.py: WORLD_SIZE %d does not match --gpus %d" % (world, args.gpus))
	ensure_built(args.no_build, local_rank)

	import numpy as np
	import torch
	import torch.distributed as dist

	setup_only = args.setup_only
	if not setup_only and not torch.cuda.is_available():
		raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
	device = None
	if not setup_only:
		torch.cuda.set_device(local_rank)
		device = torch.device("cuda", local_rank)
	# PVC3_BENCH_FORCE_DIST=1: go through RCCL (process group, barriers, the three all-reduces) with a single rank too, so that
	# the N > 1 code path can be exercised on a one-GPU box (tests/test_multiproc_gpu.py).
	use_dist = world > 1 or "1" == os.environ.get("PVC3_BENCH_FORCE_DIST")
	if use_dist:
		os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
		os.environ.setdefault("MASTER_PORT", "29533")
		if setup_only:
			dist.init_process_group(backend="gloo", rank=rank, world_size=world)
		else:
			dist.init_process_group(backend="nccl", device_id=device, rank=rank, world_size=world)		# "nccl" is RCCL on ROCm

	import panvc3_amd
	from panvc3_amd import build as pvc3_build
	from panvc3_amd.shard import reduce_job
	from tests.synth import synth

	def barrier():
		if use_dist:
			dist.barrier()

	cores = usable_cpus()
	threads = max(1, cores // max(1, world))
	config = args.config.upper()

	# ---- synthetic workload ---------------------------------------------------------------------------------------------
	gp, rp, desc = synth.config(config, scale=args.scale)
	if "C4" == config:
		# C4 is 200 M pairs over 8 GPUs: every rank generates and owns its own shard of 50 M records (never the 400 M whole).
		rp.n_records = max(2, rp.n_records // 8)
		desc += "; this rank's shard: %d records" % rp.n_records
	rp.seed = rp.seed + 1000 * rank			# every rank samples its own reads (weak scaling); the genome is identical
	t0 = time.time()
	if world > 1:
		# One genome for the node: rank 0 generates it with every granted core into a file under /dev/shm, all ranks map that
		# file (one copy of its pages, however many ranks) and the name is removed once everybody has it.  (All ranks of a
		# single-node job see the same /dev/shm; the driver runs one node.)
		shared = "/dev/shm/pvc3_bench_%s_%g_%s.genome" % (config, args.scale, os.environ.get("MASTER_PORT", "0"))
		if 0 == rank:
			g0 = synth.Genome(gp, n_threads=min(cores, gp.n_contigs))
			g0.save(shared + ".tmp")
			os.replace(shared + ".tmp", shared)
			del g0
		barrier()
		genome = synth.Genome(mapped_from=shared)
		barrier()
		if 0 == rank:
			os.unlink(shared)
	else:
		genome = synth.Genome(gp, n_threads=min(threads, gp.n_contigs))
	t1 = time.time()
	reads = synth.Reads(genome, rp, n_threads=threads)
	batch = reads.batch()
	n_local = int(reads.n)
	if args.sorted:
		order = np.lexsort((batch["src_pos"], batch["src_entry"]))
		batch = synth.reorder_batch(batch, order)
	t_reads = time.time()
	log("%s; scale %.4g: genome %.1f s, %d records %.1f s (%d host threads)" % (desc, args.scale, t1 - t0, n_local, t_reads - t1, threads))

	index = panvc3_amd.Index()
	genome.fill_index(index)
	import resource
	# every rank's set-up on one line each (stderr): what an N-rank launch costs before the first kernel
	print("[bench] rank %d of %d: set-up %.1f s (genome %.1f, reads %.1f, index on the host %.1f; %d threads), peak resident set %.0f MB"
		  % (rank, world, time.time() - t0, t1 - t0, t_reads - t1, time.time() - t_reads, threads, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0), file=sys.stderr, flush=True)
	if setup_only:
		barrier()
		if use_dist:
			dist.destroy_process_group()
		if 0 == rank:
			print(json.dumps({"setup_only": True, "n_ranks": world, "records_per_rank": n_local, "seconds": time.time() - t0}), flush=True)
		return
	ctx = panvc3_amd.Context(local_rank)
	ctx.set_option("window_read_length", int(rp.read_len))		# position windows of 128 bytes up to 158 bases, 256 beyond (what a caller that knows its reads passes)
	for kv in args.option:
		k, v = kv.split("=", 1)
		ctx.set_option(k, int(v))
	t_up = time.time()
	ctx.upload_index(index)
	t_up = time.time() - t_up
	ctx.set_timing(True)
	t2 = time.time()
	db = panvc3_amd.DeviceBatch(ctx, batch)
	index_bytes, _ = ctx.hbm_bytes()
	log("index %.2f GB in HBM (replicated with its paired blocks and position windows in %.2f s); batch uploaded in %.1f s" % (index_bytes / 1e9, t_up, time.time() - t2))
	params = panvc3_amd.api.make_params(0, -1, args.score_mode)
	if world > 1:
		# The host copies are only needed by the single-GPU legs (CPU baseline, host-buffer entry points): with one rank per
		# GPU on one node they would add up to > 20 GB of host memory per rank.
		batch = None
		del reads, genome, index

	# ---- warm-up (also sizes the workspaces) -------------------------------------------------------------------------------
	stats = None
	for _ in range(max(1, args.warmup)):
		db.project(params)
		stats = db.finish()
	if stats["n_failed"]:
		raise SystemExit("unexpected: %d records failed" % stats["n_failed"])

	# ---- timed region ------------------------------------------------------------------------------------------------------
	ms_keys = ("project", "mid", "general", "realign", "pack")
	ms = dict.fromkeys(ms_keys, 0.0)
	barrier()
	torch.cuda.synchronize()
	t_begin = time.perf_counter()
	for _ in range(args.steps):
		db.project(params)
		st = db.finish()		# waits for the step on the kernels' stream and reads the hipEvent timings
		for k in ms:
			ms[k] += st["ms_" + k]
	torch.cuda.synchronize()
	barrier()
	elapsed = time.perf_counter() - t_begin
	checksum = db.checksum(first_record=rank * n_local)
	elapsed_max, n_total, checksum_total = reduce_job(dist if use_dist else None, device, elapsed, n_local, checksum, force=use_dist)

	result = None
	if rank == 0:
		steps = args.steps
		ms_avg = {k: v / steps for k, v in ms.items()}
		ms_step = elapsed_max / steps * 1e3
		value = n_total * steps / elapsed_max
		project_bytes = stats["algorithmic_bytes_project"]
		achieved = project_bytes / (ms_avg["project"] * 1e-3) / 1e9 if ms_avg["project"] > 0 else 0.0
		traffic, traffic_src = traffic_for(config, args.scale)
		result = {
			"metric": METRIC.get(config, METRIC["C3"]),
			"value": value, "unit": "alignments/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
			"ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
			"dtype": "u64 bit-vectors / u32 CIGAR (f64 only in the realign DP)", "data": "synthetic",
			"config": {
				"workload": desc, "records_per_gpu": n_local, "read_len": int(rp.read_len), "paired": bool(rp.paired),
				"founders": int(gp.n_founders), "genome_bp": int(gp.genome_len), "scale": args.scale, "record_order": "coordinate-sorted" if args.sorted else "as sampled (uniformly random positions)", "parallelism": "records sharded, index replicated, no collective",
			},
			"roofline": {
				"bound": "hbm", "kernel": "k_project_fast", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
				"traffic": traffic, "traffic_source": (traffic_src or {}).get("source"), "algorithmic_bytes_per_launch": project_bytes, "ms_per_launch": ms_avg["project"],
				"bytes_per_alignment": project_bytes / max(1, n_local),
				"whole_step": {"algorithmic_bytes": stats["algorithmic_bytes"], "GBps": stats["algorithmic_bytes"] / (ms_step * 1e-3) / 1e9, "frac": stats["algorithmic_bytes"] / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBPS},
			},
			"kernel_ms": ms_avg, "kernel_key": pvc3_build.kernel_key()[:16],
			"batch": {"fast_path_records": stats["n_fast_path"], "mid_path_records": stats["n_mid_path"], "realigned_records": stats["n_realigned_records"], "indel_runs": stats["n_runs"], "out_cigar_ops": stats["out_cigar_total"], "checksum": "%016x" % checksum_total},
		}

		# ---- the same records in coordinate order (secondary data point; `value` above is the unsorted worst case) ------------
		# Both workflows run `samtools sort` before project_alignments (test-workflow/Snakefile:66-86): its real input is ordered
		# by (RNAME, POS), which is kinder to the caches than the uniformly random order of the main measurement.
		if world == 1 and not args.sorted and not args.skip_sorted_leg and not args.skip_host_buffers:
			order = np.lexsort((batch["src_pos"], batch["src_entry"]))
			sbatch = synth.reorder_batch(batch, order)
			del order
			db2 = panvc3_amd.DeviceBatch(ctx, sbatch)
			del sbatch
			for _ in range(2):
				db2.project(params)
				st2 = db2.finish()
			ms2 = dict.fromkeys(ms_keys, 0.0)
			torch.cuda.synchronize()
			tb = time.perf_counter()
			for _ in range(steps):
				db2.project(params)
				st2 = db2.finish()
				for k in ms2:
					ms2[k] += st2["ms_" + k]
			torch.cuda.synchronize()
			dt = time.perf_counter() - tb
			db2.close()
			fast_ms = ms2["project"] / steps
			result["sorted_input"] = {
				"what": "the same records ordered by (source sequence, position) like a coordinate-sorted BAM; mates are then not neighbours, so each record looks its mate up itself (+132 algorithmic bytes)",
				"value": n_local * steps / dt, "unit": "alignments/s", "ms_per_step": dt / steps * 1e3, "kernel_ms": {k: v / steps for k, v in ms2.items()},
				"roofline": {"kernel": "k_project_fast", "algorithmic_bytes_per_launch": st2["algorithmic_bytes_project"], "ms_per_launch": fast_ms,
							 "achieved": st2["algorithmic_bytes_project"] / (fast_ms * 1e-3) / 1e9, "frac": st2["algorithmic_bytes_project"] / (fast_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS},
			}

		# ---- host-buffer entry point (PCIe-inclusive; never `value`): upload + kernels + download of a slice, no overlap ----
		if world == 1 and not args.skip_host_buffers:
			k = int(min(n_local, 4_000_000))
			sb = synth.slice_batch(batch, 0, k)
			from panvc3_amd import api as pvc3_api
			_, _ = pvc3_api.batch_project_call(ctx, synth.slice_batch(batch, 0, min(k, 100_000)), params)		# first-use allocations
			# pvc3_batch_project itself (the call INTEGRATION.md §2 shows): ordinary arrays in, ordinary arrays out; three calls in a row, the
			# result arrays of the first used again by the others (a caller in a loop has them)
			rates, keep_out = [], None
			for _ in range(3):
				tb = time.perf_counter()
				res_hb, _ = pvc3_api.batch_project_call(ctx, sb, params, out=keep_out)
				rates.append(k / (time.perf_counter() - tb))
				keep_out = res_hb["_arrays"]
			del res_hb, keep_out
			result["host_buffers"] = {"value": rates[-1], "unit": "alignments/s", "records": k, "calls": rates,
				"what": "pvc3_batch_project: ordinary (pageable) host arrays in, host arrays out (H2D + kernels + D2H, no overlap between batches; the arrays travel through the context's page-locked staging ring); three calls in a row, `value` is the last"}
			# the same with the arrays in page-locked memory (pvc3_host_alloc), as a caller that fills them directly would have them
			pb, owners = panvc3_amd.pinned_batch(sb)
			res0, _ = panvc3_amd.project_batch(ctx, pb, pinned_out=True)		# the output arrays are reused below
			owner0 = res0.pop("_owner")
			tb = time.perf_counter()
			panvc3_amd.project_batch(ctx, pb, out=res0)
			dt = time.perf_counter() - tb
			result["host_buffers"]["pinned"] = {"value": k / dt, "unit": "alignments/s", "what": "the same batch from and to page-locked arrays"}
			del pb, owners, res0, owner0
			if hasattr(panvc3_amd, "pipeline_bench"):
				result["pipeline"] = panvc3_amd.pipeline_bench(ctx, batch, params)

		# ---- end to end through the command line (protocol iii; never `value`): BAM file -> panvc3_project_alignments -> BAM file ----
		if world == 1 and not args.skip_host_buffers and args.cli_records > 0:
			try:
				result["cli_end_to_end"] = cli_end_to_end(args.cli_records)
			except Exception as exc:		# (a leg of its own: it must not take the bench line with it)
				result["cli_end_to_end"] = {"error": str(exc)[-300:]}

		# ---- CPU baseline: the oracle (a port of the reference algorithm) on a bounded sample, same box, rank 0 only ----
		if world == 1 and args.cpu_baseline_seconds > 0:
			from oracle import oracle as orc
			o = orc.Oracle()
			genome.fill_index(o)
			# one thread first (the reference's record loop is serial: project_alignments.cc:354, 608-611)
			probe = min(n_local, 20000)
			tb = time.perf_counter()
			o.batch_project(synth.slice_batch(batch, 0, probe), 0, -1, args.score_mode, n_threads=1)
			rate1 = probe / max(1e-6, time.perf_counter() - tb)
			sample1 = int(min(n_local, max(probe, rate1 * min(5.0, args.cpu_baseline_seconds))))
			tb = time.perf_counter()
			o.batch_project(synth.slice_batch(batch, 0, sample1), 0, -1, args.score_mode, n_threads=1)
			one_thread = {"value": sample1 / (time.perf_counter() - tb), "unit": "alignments/s", "cores": 1, "sample": "first %d records, one thread" % sample1}
			tb = time.perf_counter()
			o.batch_project(synth.slice_batch(batch, 0, probe), 0, -1, args.score_mode, n_threads=cores)
			rate = probe / max(1e-6, time.perf_counter() - tb)
			sample = int(min(n_local, max(probe, rate * args.cpu_baseline_seconds)))
			sb = synth.slice_batch(batch, 0, sample)
			tb = time.perf_counter()
			want = o.batch_project(sb, 0, -1, args.score_mode, n_threads=cores)
			cpu_s = time.perf_counter() - tb
			result["cpu_baseline"] = {
				"value": sample / cpu_s, "unit": "alignments/s", "cores": cores, "kind": "port",
				"sample": "first %d records of the same batch through oracle/pvc3_oracle.c on %d threads (a restatement of the reference algorithm on plain bit-vectors; the reference itself cannot be built here)" % (sample, cores),
				"one_thread": one_thread,
			}
			# parity spot check of the sample against the device results
			got = db.download()
			n_c = int(want["out_cigar_off"][sample])
			same = (
				np.array_equal(got["dst_pos"][:sample], want["dst_pos"]) and np.array_equal(got["mate_dst_pos"][:sample], want["mate_dst_pos"])
				and np.array_equal(got["out_cigar_off"][:sample + 1], want["out_cigar_off"]) and np.array_equal(got["out_cigar"][:n_c], want["out_cigar"])
				and np.array_equal(got["status"][:sample], want["status"])
				and np.array_equal(got["run_off"][:sample + 1], want["run_off"]) and np.array_equal(got["runs"][:int(want["run_off"][sample])], want["runs"])
			)
			result["parity_sample"] = {"records": sample, "bit_identical": bool(same)}
			if not same:
				raise SystemExit("parity failure against the oracle on the bench sample")

	db.close()
	ctx.close()
	if rank == 0:
		# ---- the other single-GPU configurations of BASELINE.json at full size (never `value`): C5 = configs[4], C3G = C3 on a gap-dense MSA ----
		if world == 1 and not args.skip_host_buffers and args.other_configs:
			batch = reads = genome = index = None
			result["other_configs"] = {}
			specs, cur = [], None
			for piece in [c.strip() for c in args.other_configs.split(",") if c.strip()]:		# "C5", "C3:position_windows=0" (further options of the same configuration follow as key=value)
				if "=" in piece and ":" not in piece and cur is not None:
					cur[1].append(tuple(piece.split("=", 1)))
					continue
				name, _, opt = piece.partition(":")
				cur = [name.upper(), [tuple(opt.split("=", 1))] if opt else []]
				specs.append(cur)
			for name, options in specs:
				oc = name + "".join(":%s=%s" % kv for kv in options)
				if name == config and not options:
					continue
				try:
					result["other_configs"][oc] = measure_other_config(name, args, local_rank, threads, cores, options)
					log("%s: %.2f ms per step" % (oc, result["other_configs"][oc]["ms_per_step"]))
				except SystemExit:
					raise
				except Exception as exc:
					result["other_configs"][oc] = {"error": str(exc)[-300:]}
		print(json.dumps(result), flush=True)
	if use_dist:
		dist.barrier()
		dist.destroy_process_group()


if __name__ == "__main__":
	main()
