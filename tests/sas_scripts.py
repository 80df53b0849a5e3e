"""The transport setup of the SAS golden vectors lives in the package (roger_amd/models/svat_transport.py: the same class
runs through the reference, `make_transport_model("roger", ...)`, in tests/golden/make_golden_sas.py and through the hip
backend in the tests)."""
from roger_amd.models.svat_transport import make_transport_model  # noqa: F401
