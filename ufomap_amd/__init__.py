"""ufomap_amd -- MI355X-native scan-integration path for UFOMap (hot path only)."""
from . import scans  # noqa: F401


def __getattr__(name):
    if name in ("OccupancyMap", "OccupancyMapColor", "PointCloud", "PointCloudColor", "Comm"):
        from . import occupancy_map
        return getattr(occupancy_map, name)
    raise AttributeError(name)
