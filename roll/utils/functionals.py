"""The functions of roll/utils/functionals.py that sit on the infer path, under their reference names
(pad_to_length :351-361, get_pad_mask :301-313, postprocess_generate :768-872, GenerateRequestType :761-766,
get_dist_info_from_comm_plan :875-882)."""
import enum

from socioreasoner_amd.hostops import (concatenate_input_and_output, gather_unpadded_input_ids, get_dist_info_from_comm_plan,  # noqa: F401
                                       get_pad_mask, pad_to_length, postprocess_generate)


class GenerateRequestType(enum.Enum):
    ADD = enum.auto()
    ABORT = enum.auto()
    STOP = enum.auto()
    ALIVE_CHECK = enum.auto()
