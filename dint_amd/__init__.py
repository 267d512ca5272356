"""dint_amd — MI355X-native DINT (dictionary-of-integer-sequences) decode path.

    dint_amd.host    offline CPU half: synthetic collections, DSF dictionaries, vroom encoder
    dint_amd.device  the HIP decode path behind the C ABI of include/dint_hip.h

The device module needs dint_amd/libdint_hip.so (built by __graft_entry__.build());
there is no CPU fallback for decoding.
"""
__all__ = ["host", "device"]
