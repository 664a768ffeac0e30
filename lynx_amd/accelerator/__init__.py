from .bpm import BPM, Marker  # noqa: F401
from .cavity import Cavity  # noqa: F401
from .corrector import HorizontalCorrector, VerticalCorrector  # noqa: F401
from .custom_transfer_map import CustomTransferMap  # noqa: F401
from .dipole import Dipole, RBend  # noqa: F401
from .drift import Drift  # noqa: F401
from .element import Element  # noqa: F401
from .quadrupole import Quadrupole  # noqa: F401
from .aperture import Aperture  # noqa: F401
from .screen import Screen  # noqa: F401
from .segment import Segment  # noqa: F401
from .solenoid import Solenoid, Undulator  # noqa: F401
