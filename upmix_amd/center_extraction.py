"""Drop-in alias: ``import upmix_amd.center_extraction as ce`` exposes the names main.py uses on the reference module."""
from .plan import *          # noqa: F401,F403
from .plan import EPS        # noqa: F401
from .extractor import (MultiBandExtractorAccu, chain_bands,   # noqa: F401
                        extract_center_left_right_multi_band_in_memory)
