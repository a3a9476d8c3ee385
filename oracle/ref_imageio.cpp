// ref_imageio.cpp — TEST INFRASTRUCTURE: the reference's own image decoders, compiled where they lie.
//
// The reference reads textures with stb_image, environment maps with tinyexr and writes its pictures with
// stb_image_write / tinyexr (src/imageio.cpp:1-9).  Those three are vendored single-header libraries
// (/root/reference/include/stb/stb_image.h, stb_image_write.h and /root/reference/src/tinyexr.h) that compile on
// their own with g++, so — unlike the renderer, which needs the CUDA runtime — this part of the reference CAN be
// built here.  This file adds nothing but the calls src/imageio.cpp makes, behind a C ABI that tests/ can load:
//
//   ref_stbi_load_flipped   stbi_set_flip_vertically_on_load(true); stbi_load(file, &w, &h, &comp, 0)   imageio.cpp:13-14
//   ref_stbi_write_png      stbi_write_png(file, w, h, 3, data, 0)                                      imageio.cpp:74
//   ref_load_exr            LoadEXR(&out, &w, &h, file, &err)                                           imageio.cpp:84
//   ref_save_exr            SaveEXRImageToFile with the header ImageIO::SaveExr fills in                imageio.cpp:104-161
//                           (B, G, R channels; float in memory; HALF or FLOAT in the file), with the compression
//                           selectable so that the tests can make NONE / RLE / ZIPS / ZIP / PIZ files with the
//                           reference's own encoder
//
// src/imageio.cpp itself is not compiled: it includes <stb\stb_image.h> (a back-slash path) and imageio.h needs the
// CUDA vector types.  Its wrapper arithmetic (1/255, powf(x, 2.2f), the row flip of SavePng, Texture::Texture's
// truncation) is a few lines restated in the tests beside the calls.
//
// Built by oracle/Makefile into oracle/_ref/libref_imageio.so (git-ignored) when /root/reference is present.  Only
// tests/ may load it; nothing in the product links or calls it.
#include <cstdlib>
#include <cstring>

#define STB_IMAGE_IMPLEMENTATION
#include <stb/stb_image.h>
#define STB_IMAGE_WRITE_IMPLEMENTATION
#include <stb/stb_image_write.h>
#define TINYEXR_IMPLEMENTATION
#include <tinyexr.h>

#define REF_API extern "C" __attribute__((visibility("default")))

REF_API unsigned char *ref_stbi_load_flipped(const char *path, int *width, int *height, int *components)
{
    stbi_set_flip_vertically_on_load(true);
    return stbi_load(path, width, height, components, 0);
}

REF_API void ref_free(void *p) { free(p); }

REF_API int ref_stbi_write_png(const char *path, int width, int height, const unsigned char *rgb)
{
    return stbi_write_png(path, width, height, 3, rgb, 0);
}

// rgba: width * height * 4 floats, row 0 = top, released with ref_free
REF_API int ref_load_exr(const char *path, float **rgba, int *width, int *height)
{
    const char *err = nullptr;
    int ret = LoadEXR(rgba, width, height, path, &err);
    if (ret != TINYEXR_SUCCESS && err) FreeEXRErrorMessage(err);
    return ret;
}

// planes r, g, b: width * height floats each, row 0 = top.  compression: TINYEXR_COMPRESSIONTYPE_* (0 NONE, 1 RLE,
// 2 ZIPS, 3 ZIP, 4 PIZ); file_half: 1 = HALF in the file (what SaveExr asks for), 0 = FLOAT.
REF_API int ref_save_exr(const char *path, int width, int height, const float *r, const float *g, const float *b,
                         int compression, int file_half)
{
    EXRHeader header;
    InitEXRHeader(&header);
    EXRImage image;
    InitEXRImage(&image);
    image.num_channels = 3;
    const float *planes[3] = {b, g, r};
    image.images = (unsigned char **)planes;
    image.width = width;
    image.height = height;
    header.num_channels = 3;
    header.compression_type = compression;
    header.channels = (EXRChannelInfo *)malloc(sizeof(EXRChannelInfo) * 3);
    memset(header.channels, 0, sizeof(EXRChannelInfo) * 3);
    strcpy(header.channels[0].name, "B");
    strcpy(header.channels[1].name, "G");
    strcpy(header.channels[2].name, "R");
    header.pixel_types = (int *)malloc(sizeof(int) * 3);
    header.requested_pixel_types = (int *)malloc(sizeof(int) * 3);
    for (int i = 0; i < 3; i++) {
        header.pixel_types[i] = TINYEXR_PIXELTYPE_FLOAT;
        header.requested_pixel_types[i] = file_half ? TINYEXR_PIXELTYPE_HALF : TINYEXR_PIXELTYPE_FLOAT;
    }
    const char *err = nullptr;
    int ret = SaveEXRImageToFile(&image, &header, path, &err);
    if (ret != TINYEXR_SUCCESS && err) FreeEXRErrorMessage(err);
    free(header.channels);
    free(header.pixel_types);
    free(header.requested_pixel_types);
    return ret;
}
