"""The ReCoDe file header (v0.2: 512 bytes, 31 little-endian fields; v0.1: 321 bytes, read-only legacy).
Same interface as reference pyrecode/recode_header.py; field tables from :27-56 (v0.1) and :58-94 (v0.2),
serialisation rules from :257-275 (names space-padded UTF-8, byte arrays raw, integers little-endian)."""
import numpy as np

from .misc import get_dtype_code, get_dtype_string  # noqa: F401  (re-exported like the reference module)

UID = 158966344846346

# name, bytes, numpy type.  Multi-byte uint8 fields are byte strings / arrays.
_V02 = (
    ("uid", 8, np.uint64), ("version_major", 1, np.uint8), ("version_minor", 1, np.uint8),
    ("is_intermediate", 1, np.uint8), ("reduction_level", 1, np.uint8), ("rc_operation_mode", 1, np.uint8),
    ("is_bit_packed", 1, np.uint8), ("target_bit_depth", 1, np.uint8), ("nx", 4, np.uint32), ("ny", 4, np.uint32),
    ("nz", 4, np.uint32), ("frame_metadata_size", 1, np.uint8), ("num_non_standard_frame_metadata", 1, np.uint8),
    ("L2_statistics", 1, np.uint8), ("L4_centroiding", 1, np.uint8), ("compression_scheme", 1, np.uint8),
    ("compression_level", 1, np.uint8), ("source_file_type", 1, np.uint8), ("source_header_length", 2, np.uint16),
    ("source_header_position", 1, np.uint8), ("source_file_name", 100, np.uint8), ("calibration_file_name", 100, np.uint8),
    ("calibration_threshold_epsilon", 8, np.uint64), ("has_calibration_data", 1, np.uint8), ("frame_offset", 4, np.uint32),
    ("calibration_frame_offset", 4, np.uint32), ("num_calibration_frames", 4, np.uint32), ("source_bit_depth", 1, np.uint8),
    ("source_dtype", 1, np.uint8), ("target_dtype", 1, np.uint8), ("checksum", 32, np.uint8), ("futures", 219, np.uint8),
)
_V01 = (
    ("uid", 8, np.uint64), ("version_major", 1, np.uint8), ("version_minor", 1, np.uint8), ("reduction_level", 1, np.uint8),
    ("rc_operation_mode", 1, np.uint8), ("target_bit_depth", 1, np.uint8), ("nx", 2, np.uint16), ("ny", 2, np.uint16),
    ("nz", 4, np.uint32), ("L2_statistics", 1, np.uint8), ("L4_centroiding", 1, np.uint8), ("compression_scheme", 1, np.uint8),
    ("compression_level", 1, np.uint8), ("source_file_type", 1, np.uint8), ("source_header_length", 2, np.uint16),
    ("source_header_position", 1, np.uint8), ("source_file_name", 100, np.uint8), ("calibration_file_name", 100, np.uint8),
    ("calibration_threshold_epsilon", 2, np.uint16), ("has_calibration_data", 1, np.uint8), ("frame_offset", 4, np.uint32),
    ("calibration_frame_offset", 4, np.uint32), ("num_calibration_frames", 4, np.uint32), ("source_bit_depth", 1, np.uint8),
    ("source_dtype", 1, np.uint8), ("target_dtype", 1, np.uint8), ("checksum", 32, np.uint8), ("futures", 42, np.uint8),
)
_NAME_FIELDS = ("source_file_name", "calibration_file_name")


class ReCoDeHeader:

    def __init__(self, version=0.2):
        self._version = version
        self._rc_header = {}
        self._source_header = None
        self._non_standard_frame_metadata_sizes = {}
        self._get_rc_field_defs()

    def _get_rc_field_defs(self):
        table = _V01 if self._version < 0.2 else _V02
        self._rc_header_field_defs = [{"name": n, "bytes": b, "dtype": t} for n, b, t in table]
        self._rc_header_length = sum(b for _, b, _ in table)  # 321 / 512

    def create(self, init_params, input_params, is_intermediate):
        ip, h = input_params, {}
        h["uid"] = UID
        h["version_major"] = 0
        h["version_minor"] = 1 if self._version < 0.2 else 2
        for name in ("reduction_level", "rc_operation_mode", "target_bit_depth", "nx", "ny", "nz", "L2_statistics",
                     "L4_centroiding", "compression_scheme", "compression_level", "source_file_type",
                     "source_header_length", "calibration_threshold_epsilon", "frame_offset",
                     "calibration_frame_offset", "num_calibration_frames", "source_bit_depth"):
            h[name] = getattr(ip, name)
        h["source_header_position"] = 0
        h["source_file_name"] = init_params.image_filename
        h["calibration_file_name"] = init_params.calibration_filename
        h["has_calibration_data"] = ip.keep_calibration_data
        if self._version < 0.2:  # v0.1 knows unsigned integers only
            h["source_dtype"] = h["target_dtype"] = 0
        else:
            h["is_intermediate"] = is_intermediate
            h["is_bit_packed"] = 1
            h["frame_metadata_size"] = 0
            h["num_non_standard_frame_metadata"] = 0
            h["source_dtype"], h["target_dtype"] = ip.source_data_type, ip.target_data_type
        for d in self._rc_header_field_defs:
            if d["name"] in ("checksum", "futures"):
                h[d["name"]] = np.zeros(d["bytes"], dtype=np.uint8)
        self._rc_header = h

    @property
    def recode_header_length(self):
        return self._rc_header_length

    def as_dict(self):
        return self._rc_header

    def get(self, field_name):
        if field_name not in self._rc_header:
            raise ValueError("The requested field does not exist in recode header")
        return self._rc_header[field_name]

    def get_definition(self, name):
        for d in self._rc_header_field_defs:
            if d["name"] == name:
                return d
        raise ValueError("The requested field does not exist in recode header")

    def set(self, field_name, value):
        if field_name not in self._rc_header:
            raise ValueError("The requested field does not exist in recode header")
        self._rc_header[field_name] = value

    def update(self, name, value):
        self._rc_header[name] = value

    def load(self, rc_filename, is_intermediate=False):
        if rc_filename == "":
            raise ValueError("ReCoDe filename missing")
        with open(rc_filename, "rb") as fp:
            head = fp.read(10)
            major, minor = head[8], head[9]
            self._version = int(major) + int(minor) / 10.0
            self._get_rc_field_defs()
            fp.seek(0)
            blob = fp.read(self._rc_header_length)
            pos = 0
            for d in self._rc_header_field_defs:
                raw = blob[pos:pos + d["bytes"]]
                pos += d["bytes"]
                if d["name"] in _NAME_FIELDS:
                    value = raw.decode("latin-1")
                else:
                    arr = np.frombuffer(raw, dtype=d["dtype"])
                    value = arr[0] if arr.size == 1 else arr
                self._rc_header[d["name"]] = value
            if self._version < 0.2:
                self._rc_header.update(is_intermediate=0 if is_intermediate else 1, is_bit_packed=1, frame_metadata_size=0,
                                       num_non_standard_frame_metadata=0, source_header_length=0, source_dtype=0,
                                       target_dtype=0)
            for _ in range(int(self._rc_header["num_non_standard_frame_metadata"])):
                entry = fp.read(100)
                self._non_standard_frame_metadata_sizes[entry[:99].decode("latin-1")] = entry[99]
            self._source_header = fp.read(int(self._rc_header["source_header_length"]))

    def serialize(self, rc_filename):
        if rc_filename == "":
            raise ValueError("ReCoDe filename missing")
        with open(rc_filename, "wb") as fp:
            self.serialize_to(fp)

    def to_bytes(self):
        out = bytearray()
        for d in self._rc_header_field_defs:
            name, size, value = d["name"], d["bytes"], self._rc_header[d["name"]]
            if name in _NAME_FIELDS:
                out += str(value)[:size].ljust(size, " ").encode("utf-8")[:size]
            elif d["dtype"] == np.uint8 and size != 1:
                out += np.asarray(value, dtype=np.uint8)[:size].tobytes()
            else:
                out += int(value).to_bytes(size, "little")
        return bytes(out)

    def serialize_to(self, fp):
        fp.write(self.to_bytes())

    def skip_header(self, rc_fp):
        rc_fp.seek(self._rc_header_length)
        return rc_fp

    def get_frame_data_offset(self, is_intermediate, sz_frame_metadata):
        h = self._rc_header
        offset = self._rc_header_length
        if not (h["version_major"] == 0 and h["version_minor"] == 1):
            offset += int(h["source_header_length"]) + 100 * len(self._non_standard_frame_metadata_sizes)
        return offset if is_intermediate else int(offset + int(h["nz"]) * sz_frame_metadata)

    @property
    def source_header(self):
        return self._source_header

    @property
    def non_standard_metadata_sizes(self):
        return self._non_standard_frame_metadata_sizes

    def get_field_position_in_bytes(self, name):
        position = 0
        for d in self._rc_header_field_defs:
            if d["name"] == name:
                return position
            position += d["bytes"]
        raise ValueError("The requested field is not defined in the header")

    def print(self):
        print("ReCoDe Header")
        print("-------------")
        for d in self._rc_header_field_defs:
            print(d["name"], "=", self._rc_header[d["name"]])

    def validate(self):
        for d in self._rc_header_field_defs:
            if d["name"] not in self._rc_header:
                print("ReCoDe Header Validation Failed: " + d["name"] + " is missing.")
                return False
        return True
