"""The division-by-zero corner of the reference's D' (linkage.py:187-195).

The reference divides Python floats: `linkd / max(-fA*fB, -fa*fb)` when linkd < 0, `linkd / min(fA*fb, fa*fB)` when
linkD > 0.  A zero denominator needs one allele of one site absent among the reads that span both sites; then linkd and
linkD are 0 in exact arithmetic and neither branch is taken -- a ZeroDivisionError (= a failed split,
profile_utilities.py:104-111) would need the fp64 rounding of the frequencies to push linkd below or linkD above zero.
This test replays the reference's arithmetic, operation by operation in Python floats, over every count table that can
reach the division (total > min_snp = 20) up to a bound, and finds no such table: the product's "nan D' on a
monomorphic pair table" is the reference's behaviour, not a deviation.  (The search run once for the write-up went
further: every table up to total 119, every table with an absent allele up to total 3000.)"""


def d_prime_like_the_reference(AB, Ab, aB, ab):
    total = AB + Ab + aB + ab
    t = float(total)
    fAB, fAb, faB, fab = AB / t, Ab / t, aB / t, ab / t
    fA, fa, fB, fb = fAB + fAb, fab + faB, fAB + faB, fab + fAb
    linkD = fAB - fA * fB
    linkd = fab - fa * fb
    if linkd < 0:
        return linkd / max([-fA * fB, -fa * fb])          # ZeroDivisionError propagates
    if linkD > 0:
        return linkd / min([fA * fb, fa * fB])
    return float("nan")


def test_no_count_table_reaches_the_division_by_zero():
    n = 0
    for total in range(21, 49):                              # every table
        for AB in range(total + 1):
            for Ab in range(total + 1 - AB):
                for aB in range(total + 1 - AB - Ab):
                    d_prime_like_the_reference(AB, Ab, aB, total - AB - Ab - aB)
                    n += 1
    for total in range(49, 700):                             # every table with one allele of one site absent
        for x in range(total + 1):
            y = total - x
            for tab in ((0, 0, x, y), (x, y, 0, 0), (0, x, 0, y), (x, 0, y, 0)):
                d_prime_like_the_reference(*tab)
                n += 1
    assert n > 500000


def test_monomorphic_pair_table_gives_nan_in_the_oracle():
    """what the oracle (and the product, tests/test_gpu_linkage*.py) return where a zero denominator would sit"""
    import math
    from oracle import py_columns  # noqa: F401  (the restatement holds the same branch structure)
    for tab in ((0, 0, 12, 13), (25, 0, 5, 0), (0, 30, 0, 1)):
        assert math.isnan(d_prime_like_the_reference(*tab))
