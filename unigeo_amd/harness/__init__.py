"""Host-side mirror of the reference harness (eval.py loop, GT preparation, depth / normal metrics, CSV sink).
Plain numpy / Python; the model call inside the loop is the only GPU work."""
from .io_utils import prepare_gt_label
from .metrics import MetricsManager, depth_evaluation, normal_evaluation
from .eval import evaluate, parse_dataset_config, parse_metric_config, import_class_from_module
from .dataset import SyntheticGeometryDataset, split_clips
from .scannetpp import ScannetPPDataset, ScannetPPSequence
from .distributed import evaluate_sharded

__all__ = ["prepare_gt_label", "MetricsManager", "depth_evaluation", "normal_evaluation", "evaluate",
           "parse_dataset_config", "parse_metric_config", "import_class_from_module", "SyntheticGeometryDataset",
           "split_clips", "evaluate_sharded", "ScannetPPDataset", "ScannetPPSequence"]
