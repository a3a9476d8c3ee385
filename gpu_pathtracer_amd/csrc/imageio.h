// imageio.h — host image files (see imageio.cpp).
#pragma once
#include <vector>
#include "../../include/gpt_types.h"

namespace imageio {
bool write_png_rgb8(const char *path, int width, int height, const unsigned char *rgb_top_down);
bool read_png(const char *path, int &width, int &height, int &components, std::vector<unsigned char> &rgba_top_down);
bool read_jpeg(const char *path, int &width, int &height, int &components, std::vector<unsigned char> &rgba_top_down);
bool read_bmp(const char *path, int &width, int &height, int &components, std::vector<unsigned char> &rgba_top_down);
bool read_tga(const char *path, int &width, int &height, int &components, std::vector<unsigned char> &rgba_top_down);
bool read_pnm(const char *path, int &width, int &height, int &components, std::vector<unsigned char> &rgba_top_down);
bool read_any8(const char *path, int &width, int &height, int &components, std::vector<unsigned char> &rgba_top_down);   // JPEG, PNG, BMP, PNM, TGA by content
bool load_texture(const char *path, int &width, int &height, std::vector<gpt_uchar4> &texels);   // any of those
bool write_pfm(const char *path, int width, int height, const float *rgb_bottom_up);
bool read_pfm_top_down(const char *path, int &width, int &height, std::vector<gpt_float3> &out);
bool read_exr_rgb_top_down(const char *path, int &width, int &height, std::vector<gpt_float3> &out);
}  // namespace imageio
