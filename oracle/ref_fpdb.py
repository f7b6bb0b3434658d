"""TEST INFRASTRUCTURE ONLY (never imported by the product path): numpy restatement of the reference's fingerprint-DB
writer and reader, used by tests/test_fpdb.py to pin neuralsampleid_amd/fpdb.py byte for byte.

Reference: test_fp.py:120-133 (writer, identical in create_fp_db / create_ref_db / create_dummy_db),
eval.py:179-196 (reader `load_memmap_data`). The reference modules themselves cannot be imported here (faiss, torchaudio
are absent), so this file follows their text line by line; the format has no arithmetic, only layout."""
import json

import numpy as np


def write_db(output_root_dir, fname, fp_list, lookup_table):
    fp = np.concatenate(fp_list)                                   # test_fp.py:120
    arr_shape = (len(fp), fp_list[-1].shape[-1])                   # :121  (z_i.shape[-1])
    arr = np.memmap(f'{output_root_dir}/{fname}.mm', dtype='float32', mode='w+', shape=arr_shape)   # :123-126
    arr[:] = fp[:]                                                 # :127
    arr.flush(); del (arr)                                         # :128
    np.save(f'{output_root_dir}/{fname}_shape.npy', arr_shape)     # :130
    json.dump(lookup_table, open(f'{output_root_dir}/{fname}_lookup.json', 'w'))   # :133


def load_memmap_data(source_dir, fname, append_extra_length=None, shape_only=False):
    data_shape = np.load(f'{source_dir}/{fname}_shape.npy')        # eval.py:181
    if shape_only:
        return data_shape
    if append_extra_length:
        data_shape[0] += append_extra_length                       # :186
    data = np.memmap(f'{source_dir}/{fname}.mm', dtype='float32', mode='r+', shape=(data_shape[0], data_shape[1]))
    data[np.isnan(data)] = 0.0                                     # :193
    return data, data_shape
