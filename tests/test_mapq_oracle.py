"""The restatement of recalculate_mapq (oracle/recalculate_mapq.py) against values derived by hand from the reference's
tables (recalculate-mapq/main.cc:375-431) and formulas (:441-566, :182-289).  The reference has no test or fixture for this
tool (parity unpinned); these cases pin the restatement to the source text."""
import pytest

from helpers import ROOT  # noqa: F401  (puts the repository root on sys.path)
from oracle import recalculate_mapq as rm


def test_min_score_is_bowtie2_score_min():
	assert rm.read_min_score(0) == 0.0
	assert rm.read_min_score(150) == pytest.approx(-90.6)
	assert rm.read_min_score(100) + rm.read_min_score(100) == pytest.approx(-121.2)


@pytest.mark.parametrize("length,other,score,nxt,expected", [
	# below the minimum valid score: 0 (main.cc:490-491)
	(100, 0, -61.0, rm.ALIGNMENT_SCORE_MIN, 0),
	# a unique perfect alignment: next := min_score (:494), diff quotient 1.0 -> row (0.9, *); normalised 1.0 -> (0.9, 1.0) = 39
	(100, 0, 0.0, rm.ALIGNMENT_SCORE_MIN, 39),
	# unique, score at 50 % of the range: diff quotient 0.5 -> rows (0.5, *); normalised 0.5 -> (0.5, 0.0) = 5
	(100, 0, -30.3, rm.ALIGNMENT_SCORE_MIN, 5),
	# two equally good alignments: diff_next == 0 -> 1 if normalised >= 0.67 else 0 (:561-565)
	(100, 0, 0.0, 0.0, 1),
	(100, 0, -30.0, -30.0, 0),
	# perfect alignment, next best 6 below (one high-quality mismatch): diff quotient 6 / 60.6 = 0.099 -> rows (0.0, *); normalised 1.0 -> (0.0, 0.67) = 6
	(100, 0, 0.0, -6.0, 6),
	# next best 12 below: quotient 0.198 -> rows (0.1, *); normalised 1.0 -> (0.1, 1.0) = 30
	(100, 0, 0.0, -12.0, 30),
	# next best 30 below: quotient 0.495 -> rows (0.4, *); normalised 1.0 -> (0.4, 1.0) = 34
	(100, 0, 0.0, -30.0, 34),
	# the same with a score of -10: normalised 50.6 / 60.6 = 0.835 -> (0.3, 0.67) = 15 for quotient 20 / 60.6 = 0.33
	(100, 0, -10.0, -30.0, 15),
	# a pair of 100-base reads: range 121.2; totals 0 and -12: quotient 0.099 -> (0.0, 0.67) = 6
	(100, 100, 0.0, -12.0, 6),
	# next alignment below the minimum valid score counts as the minimum (:494): same as unique
	(100, 0, 0.0, -500.0, 39),
])
def test_calculate_mapq_hand_derived(length, other, score, nxt, expected):
	assert rm.calculate_mapq(length, other, score, nxt) == expected


def test_penalty_tables_and_cigar_score():
	mm = rm.fill_penalties(2, 6)
	assert mm[0] == 2 and mm[40] == 6 and mm[255] == 6 and mm[20] == 4 and mm[10] == 3 and mm[39] == 5		# truncation of 5.9
	# 3=1X2I1=2D1X1=: X at q3 (qual 40): -6; gap open 5 + 2 * 3; then D directly after I? no: '=' in between resets -> open 5 + 2 * 3; X at N reference: -1
	cigar = [(3, "="), (1, "X"), (2, "I"), (1, "="), (2, "D"), (1, "X"), (1, "=")]
	seq = "ACGTACGTA"
	qual = [40] * 9
	xn = bytes([0b10000000, 0b00000000])		# reference position 7 is N: the second X is at ref position 3 + 1 + 1 + 2 = 7
	assert rm.cigar_score(cigar, seq, qual, xn) == -(6 + 5 + 6 + 5 + 6 + 1)
	# adjacent I and D share one opening (:223-239)
	assert rm.cigar_score([(2, "="), (1, "I"), (1, "D"), (2, "=")], "ACGTA", [30] * 5, b"\0") == -(5 + 3 + 3)
	# N in the read under '=' costs the N penalty; soft clips and hard clips cost nothing
	assert rm.cigar_score([(1, "H"), (2, "S"), (3, "=")], "ACNGT", [30] * 5, b"") == -1
	with pytest.raises(ValueError):
		rm.cigar_score([(5, "M")], "ACGTA", [30] * 5, b"\0")
	with pytest.raises(IndexError):
		rm.cigar_score([(1, "X")], "A", [30], b"")


def _rec(ref, pos, mref, mpos, orig, orig_mate, score, seq="A" * 100):
	return dict(ref_id=ref, pos=pos, mate_ref_id=mref, mate_pos=mpos, orig=orig, orig_mate=orig_mate, score=score, seq=seq)


def test_group_of_paired_records():
	# one pair aligned to two founders that project to the same place (totals 0 and -12), and to a third place (-30)
	recs = [
		_rec(0, 1000, 0, 1300, (1, 1000), (1, 1300), 0), _rec(0, 1300, 0, 1000, (1, 1300), (1, 1000), 0),			# founder 1
		_rec(0, 1000, 0, 1300, (2, 1001), (2, 1301), -6), _rec(0, 1300, 0, 1000, (2, 1301), (2, 1001), -6),		# founder 2, same projected pair
		_rec(0, 5000, 0, 5300, (3, 5000), (3, 5300), -12), _rec(0, 5300, 0, 5000, (3, 5300), (3, 5000), -18),		# elsewhere: total -30
	]
	emitted, stats = rm.process_alignment_group(recs)
	assert emitted and stats["mate_not_found"] == 0 and stats["unpaired_alignments"] == 0
	# best pair: total 0, next best total at another position pair is -30: 30 / 121.2 = 0.2475 -> rows (0.2, *), normalised 1.0 -> 31
	assert recs[0]["mapq"] == recs[1]["mapq"] == 31
	# second founder's pair (total -12): the only other position pair with a total <= -12 is -30: 18 / 121.2 = 0.1485 -> rows (0.1, *); normalised (121.2 - 12) / 121.2 = 0.90 -> (0.1, 0.88) = 12
	assert recs[2]["mapq"] == recs[3]["mapq"] == 12
	# the worst pair: nothing at another position with a total <= -30, and better totals exist: 255
	assert recs[4]["mapq"] == recs[5]["mapq"] == rm.MAPQ_NO_NEXT_RECORD


def test_group_mixed_and_unpaired():
	mixed = [_rec(0, 10, 0, 300, (1, 10), (1, 300), 0), _rec(0, 10, None, None, (1, 10), None, 0)]
	emitted, stats = rm.process_alignment_group(mixed)
	assert not emitted and stats["reads_with_and_without_mate"] == 1
	single = [_rec(0, 10, None, None, (1, 10), None, 0), _rec(0, 900, None, None, (2, 900), None, -12), _rec(0, 10, None, None, (3, 11), None, -6)]
	emitted, stats = rm.process_alignment_group(single)
	assert emitted and stats["unpaired_alignments"] == 3
	assert single[0]["mapq"] == 30		# next best at another position: -12 -> quotient 0.198 -> (0.1, 1.0)
	assert single[1]["mapq"] == rm.MAPQ_NO_NEXT_RECORD
	# -6 at the same projected position as the best: candidates with a score <= -6 at another position: -12: 6 / 60.6 = 0.099, normalised 0.90 -> (0.0, 0.67) = 6
	assert single[2]["mapq"] == 6
	# a mate that cannot be found through YR / YP counts, and the record is rated by its own score alone
	lost = [_rec(0, 10, 0, 300, (1, 10), (1, 999), 0), _rec(0, 300, 0, 10, (1, 300), (1, 10), 0)]
	emitted, stats = rm.process_alignment_group(lost)
	assert emitted and stats["mate_not_found"] == 1
