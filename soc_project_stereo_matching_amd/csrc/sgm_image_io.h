/* sgm_image_io.h -- minimal image I/O for the command-line driver (sgm_main.c).
 * Grey conversion of colour input follows stb_image's req_comp=1 path, the one the reference's main.c uses
 * (stb_image.h:1746-1749): y = (77 r + 150 g + 29 b) >> 8. */
#ifndef SGM_IMAGE_IO_H
#define SGM_IMAGE_IO_H
#include <stdint.h>

/* Loads PGM (P5), PPM (P6) or PNG (8-bit, non-interlaced; grey, grey+alpha, RGB, RGBA, palette) as 8-bit grey.
 * Returns a malloc'ed buffer of w*h bytes or NULL (message on stderr). */
uint8_t* sgm_load_gray(const char* path, int* w, int* h);

/* 8-bit grey writers; return 0 on success. */
int sgm_write_png_gray(const char* path, const uint8_t* data, int w, int h);
int sgm_write_pgm(const char* path, const uint8_t* data, int w, int h);
#endif
