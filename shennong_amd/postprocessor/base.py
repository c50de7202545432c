"""Features --> FeaturesPostProcessor --> Features
(mirror of reference shennong/postprocessor/base.py:15-32)"""

import abc

from shennong_amd.processor.base import FeaturesProcessor
from shennong_amd.utils import copy_properties


class FeaturesPostProcessor(FeaturesProcessor):
    """Base class of all features post-processors"""
    @abc.abstractmethod
    def process(self, features):
        """Returns features post-processed from input `features`"""

    def get_properties(self, features):
        properties = copy_properties(features.properties)
        properties[self.name] = self.get_params()
        if 'pipeline' not in properties:
            properties['pipeline'] = []
        properties['pipeline'].append({
            'name': self.name,
            'columns': [0, self.ndims - 1]})
        return properties
