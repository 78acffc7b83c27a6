"""Host-side mirror of the reference's `inStrain.profile` interface for the hot path.

    inStrain.profile.profile_bam(bam, fasta_db, sR2M, ISP_loc, **kwargs)      profile/__init__.py:7-18
    inStrain.profile.profile_utilities.profile_split(...) -> SplitObject       profile_utilities.py:115-216

Same names, argument meaning and failure convention; the work is done by libinstrain_amd.so
(one batch of splits per call instead of one process per split).
"""
from .profile_utilities import SplitObject, make_coverage_table, profile_bam, profile_splits  # noqa: F401
from .snv_utilities import generate_snp_model, null_model_lut  # noqa: F401
