"""repeatafterme_amd -- MI355X-native RAMExtend extension loop."""
from .datamodel import CoreSet, FlankSet, ExtendParams, new_master  # noqa: F401
