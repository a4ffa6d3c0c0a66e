/* Test infrastructure (oracle/_ref build recipe) - NOT part of the product.
 *
 * Compiles the reference's own c_extensions/reader.h *in place* (path passed by build_ref.sh as
 * REF_READER_H; no reference source is copied into this repo) into a plain shared object so that
 * tests can call the reference's C loops directly through ctypes:
 *     _unpack_frame_sparse          (reader.h:10-68)
 *     _bit_pack_pixel_intensities   (reader.h:105-140)
 * reader.h relies on its includer for the standard headers (pyrecode.cpp:1-17 gets them through
 * Python.h), hence the three includes below. _bit_unpack_pixel_intensities (reader.h:74-99) is
 * compiled too but must never be called: its loop increments the wrong variable.
 */
#include <stdint.h>
#include <time.h>
#include <math.h>
#include REF_READER_H
