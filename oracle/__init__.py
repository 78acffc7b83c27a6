"""
oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's `inStrain profile` hot path
(/root/reference/inStrain/profile/{profile_utilities,snv_utilities,linkage}.py)
and of the third-party htslib-1.9 pileup semantics the reference reaches
through pysam (not vendored under /root/reference; `setup.py:25` pysam>=0.15).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import anything from here -- and only as the checker / the timed CPU
baseline. The product (`instrain_amd/`) never imports, links or executes
anything under `oracle/`.

Parity pinning (see DESIGN.md "Oracle"): pinned against
  * the reference's stored golden run
    test/test_data/sars_cov_2_MT039887.1.fasta.bt2-vs-SRR11140750.sam.IS
    (raw_snp_table, raw_linkage_table, cumulative_scaffold_table, read_report)
    -> committed as tests/golden/sars_cov_2_*.csv.gz
  * golden vectors produced by importing the reference's own Python
    (tests/golden/make_golden.py, run in the build container only).
"""
