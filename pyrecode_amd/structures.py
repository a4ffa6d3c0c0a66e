"""Per-frame metadata layout of .rc files for every (reduction level, operation mode).
Same interface as reference pyrecode/structures.py (fields :18-46, sizes :48-91, binary_image_sz_bytes :15-16)."""
import numpy as np

_U32 = np.uint32


def _field(name, is_size):
    return {"name": name, "bytes": 4, "dtype": _U32, "is_frame_size": is_size}


# what follows frame_id in a record / what one row of the merged file's metadata table holds
_LAYOUT = {
    (1, 0): (("bytes_in_packed_pixvals", True),),
    (1, 1): (("bytes_in_compressed_binary_map", True), ("bytes_in_compressed_pixvals", True), ("bytes_in_packed_pixvals", False)),
    (2, 0): (("bytes_in_packed_summary_stats", True),),
    (2, 1): (("bytes_in_compressed_binary_map", True), ("bytes_in_compressed_summary_stats", True),
             ("bytes_in_packed_summary_stats", False)),
    (3, 0): (), (4, 0): (),
    (3, 1): (("bytes_in_compressed_binary_map", True),),
    (4, 1): (("bytes_in_compressed_binary_map", True),),
}


class ReCoDeStructures:

    def __init__(self, recode_header):
        self._recode_header = recode_header
        n_pixels = int(recode_header["nx"]) * int(recode_header["ny"])
        self._binary_image_sz_bytes = (n_pixels + 7) // 8
        self._standard_frame_metadata_structure = {
            key: [_field(n, s) for n, s in fields] for key, fields in _LAYOUT.items()}

    def get_standard_frame_metadata_size(self, reduction_level, rc_operation_mode):
        return sum(np.dtype(f["dtype"]).itemsize for f in self._standard_frame_metadata_structure[(reduction_level, rc_operation_mode)])

    def get_frame_data_size(self, reduction_level, rc_operation_mode, metadata):
        """Bytes of frame data (metadata excluded): fixed-size raw bitmap in mode 0, the sized fields otherwise."""
        fields = self._standard_frame_metadata_structure[(reduction_level, rc_operation_mode)]
        sized = sum(int(metadata[f["name"]]) for f in fields if f["is_frame_size"])
        return sized + (self._binary_image_sz_bytes if rc_operation_mode == 0 else 0)

    @property
    def binary_image_sz_bytes(self):
        return self._binary_image_sz_bytes

    @property
    def standard_frame_metadata_structure(self):
        return self._standard_frame_metadata_structure

    def standard_frame_metadata_structure_for(self, reduction_level, rc_operation_mode):
        return self._standard_frame_metadata_structure[(reduction_level, rc_operation_mode)]
