"""Features --> FeaturesPostProcessor --> Features (counterpart of reference
shennong/postprocessor/base.py:15-32)"""

import abc

from shennong_amd.processor.base import FeaturesProcessor
from shennong_amd.utils import copy_properties


class FeaturesPostProcessor(FeaturesProcessor):
    """Base class of all features post-processors: `process` maps Features to Features and the
    properties of the result are the input's plus one pipeline stage"""
    @abc.abstractmethod
    def process(self, features):
        """Returns features post-processed from input `features`"""

    def _extend_properties(self, features, ndims):
        """A copy of the input's properties with a new stage of `ndims` columns at the end of the
        'pipeline' list"""
        properties = copy_properties(features.properties)
        properties.setdefault('pipeline', []).append(
            {'name': self.name, 'columns': [0, ndims - 1]})
        return properties

    def get_properties(self, features):
        properties = self._extend_properties(features, self.ndims)
        properties[self.name] = self.get_params()
        return properties
