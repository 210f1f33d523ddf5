"""Output records of the routed transformer forward and the pipeline call (vorta/patch/outputs.py:9-29).

The reference derives them from diffusers' `BaseOutput`; these are plain dataclasses with the same field names,
attribute access and tuple conversion, so they exist without diffusers."""
from dataclasses import dataclass, fields
from typing import Any, List, Optional, Tuple

import torch


class _Record:
    def to_tuple(self) -> Tuple[Any, ...]:
        return tuple(getattr(self, f.name) for f in fields(self) if getattr(self, f.name) is not None)

    def __getitem__(self, k):
        return getattr(self, k) if isinstance(k, str) else self.to_tuple()[k]


@dataclass
class RoutedTransformerModelOutput(_Record):
    sample: torch.Tensor
    reg_loss: Optional[torch.Tensor] = None
    last_layer_distill_loss: Optional[torch.Tensor] = None
    hidden_layer_distill_loss: Optional[torch.Tensor] = None
    routing_scores: Optional[List[torch.Tensor]] = None


@dataclass
class VideoPipelineOutput(_Record):
    """frames: tensor / ndarray / nested list of PIL images, as the stock pipeline produced them."""
    frames: Any
    routing_scores: Optional[List[List[torch.Tensor]]] = None
