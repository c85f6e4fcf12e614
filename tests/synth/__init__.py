from .synth import Genome, Reads, GenomeParams, ReadsParams, config, build  # noqa: F401
