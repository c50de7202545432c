"""MI355X-native (gfx950 / CDNA4) speech features backend

Same Python surface as bootphon/shennong for the hot path (``FeaturesProcessor.process(Audio) ->
Features``, ``process_all``, ``get_params/set_params``) with the arithmetic in hand-written HIP
kernels behind the C ABI of ``include/shennong_amd.h``.  See README.md / DESIGN.md.
"""

from shennong_amd.audio import Audio
from shennong_amd.features import Features, FeaturesCollection
from shennong_amd.utterances import Utterance, Utterances

__version__ = '0.1.0'

__all__ = ['Audio', 'Features', 'FeaturesCollection', 'Utterance', 'Utterances']
