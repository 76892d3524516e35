/* sgm_image_io.c -- PGM/PPM/PNG reader and PNG/PGM writer on top of zlib (see sgm_image_io.h). */
#include "sgm_image_io.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

static uint8_t luma(unsigned r, unsigned g, unsigned b) { return (uint8_t)((r * 77 + g * 150 + b * 29) >> 8); }

static uint8_t* read_file(const char* path, size_t* n)
{
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); return NULL; }
    long sz = -1;
    if (fseek(f, 0, SEEK_END) == 0) sz = ftell(f);
    if (sz < 0 || fseek(f, 0, SEEK_SET) != 0) { fprintf(stderr, "cannot size %s (not a regular file?)\n", path); fclose(f); return NULL; }
    uint8_t* buf = (uint8_t*)malloc(sz > 0 ? (size_t)sz : 1);
    if (!buf || fread(buf, 1, (size_t)sz, f) != (size_t)sz) { fprintf(stderr, "cannot read %s\n", path); free(buf); fclose(f); return NULL; }
    fclose(f);
    *n = (size_t)sz;
    return buf;
}

/* ---------------------------------------------------------------- PNM */

static int pnm_token(const uint8_t* b, size_t n, size_t* pos, int* value)
{
    size_t p = *pos;
    for (;;) {
        while (p < n && (b[p] == ' ' || b[p] == '\t' || b[p] == '\n' || b[p] == '\r')) ++p;
        if (p < n && b[p] == '#') { while (p < n && b[p] != '\n') ++p; continue; }
        break;
    }
    if (p >= n || b[p] < '0' || b[p] > '9') return -1;
    int v = 0;
    while (p < n && b[p] >= '0' && b[p] <= '9') v = v * 10 + (b[p++] - '0');
    *pos = p;
    *value = v;
    return 0;
}

static uint8_t* load_pnm(const uint8_t* b, size_t n, int* w, int* h)
{
    const int colour = (b[1] == '6');
    size_t pos = 2;
    int maxv;
    if (pnm_token(b, n, &pos, w) || pnm_token(b, n, &pos, h) || pnm_token(b, n, &pos, &maxv) || maxv != 255 || *w <= 0 || *h <= 0) {
        fprintf(stderr, "unsupported PNM header (need binary P5/P6 with maxval 255)\n");
        return NULL;
    }
    ++pos;                                              /* the single whitespace after maxval */
    const size_t px = (size_t)*w * *h, need = px * (colour ? 3 : 1);
    if (pos + need > n) { fprintf(stderr, "truncated PNM\n"); return NULL; }
    uint8_t* out = (uint8_t*)malloc(px);
    if (!out) return NULL;
    if (!colour) memcpy(out, b + pos, px);
    else for (size_t i = 0; i < px; ++i) out[i] = luma(b[pos + 3 * i], b[pos + 3 * i + 1], b[pos + 3 * i + 2]);
    return out;
}

/* ---------------------------------------------------------------- PNG */

static uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

static int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

static uint8_t* load_png(const uint8_t* b, size_t n, int* w, int* h)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (n < 33 || memcmp(b, sig, 8)) { fprintf(stderr, "not a PNG\n"); return NULL; }
    size_t pos = 8, idat_len = 0;
    uint8_t* idat = (uint8_t*)malloc(n);
    uint8_t palette[256][3];
    int have_ihdr = 0, depth = 0, ctype = 0, interlace = 0;
    memset(palette, 0, sizeof palette);
    while (idat && pos + 12 <= n) {
        const uint32_t len = be32(b + pos);
        const uint8_t* type = b + pos + 4;
        const uint8_t* data = b + pos + 8;
        if (pos + 12 + (size_t)len > n) break;
        if (!memcmp(type, "IHDR", 4) && len >= 13) {
            *w = (int)be32(data); *h = (int)be32(data + 4);
            depth = data[8]; ctype = data[9]; interlace = data[12];
            have_ihdr = 1;
        } else if (!memcmp(type, "PLTE", 4)) {
            for (uint32_t i = 0; i < len / 3 && i < 256; ++i) memcpy(palette[i], data + 3 * i, 3);
        } else if (!memcmp(type, "IDAT", 4)) {
            memcpy(idat + idat_len, data, len);
            idat_len += len;
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    if (!idat || !have_ihdr || depth != 8 || interlace != 0 || *w <= 0 || *h <= 0 ||
        !(ctype == 0 || ctype == 2 || ctype == 3 || ctype == 4 || ctype == 6)) {
        fprintf(stderr, "unsupported PNG (need 8-bit, non-interlaced)\n");
        free(idat);
        return NULL;
    }
    const int ch = (ctype == 0 || ctype == 3) ? 1 : (ctype == 4 ? 2 : (ctype == 2 ? 3 : 4));
    const size_t stride = (size_t)*w * ch;
    uLongf raw_len = (uLongf)((stride + 1) * (size_t)*h);
    uint8_t* raw = (uint8_t*)malloc(raw_len);
    if (!raw || uncompress(raw, &raw_len, idat, (uLong)idat_len) != Z_OK || raw_len != (stride + 1) * (size_t)*h) {
        fprintf(stderr, "PNG inflate failed\n");
        free(raw); free(idat);
        return NULL;
    }
    free(idat);
    /* undo the scanline filters in place (raw row = 1 filter byte + stride bytes) */
    for (int y = 0; y < *h; ++y) {
        uint8_t* row = raw + (size_t)y * (stride + 1) + 1;
        const uint8_t* up = y ? row - (stride + 1) : NULL;
        const int f = row[-1];
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)ch ? row[i - ch] : 0, bb = up ? up[i] : 0, c = (up && i >= (size_t)ch) ? up[i - ch] : 0;
            int add = 0;
            switch (f) {
            case 0: break;
            case 1: add = a; break;
            case 2: add = bb; break;
            case 3: add = (a + bb) >> 1; break;
            case 4: add = paeth(a, bb, c); break;
            default: fprintf(stderr, "bad PNG filter %d\n", f); free(raw); return NULL;
            }
            row[i] = (uint8_t)(row[i] + add);
        }
    }
    const size_t px = (size_t)*w * *h;
    uint8_t* out = (uint8_t*)malloc(px);
    if (out)
        for (int y = 0; y < *h; ++y) {
            const uint8_t* row = raw + (size_t)y * (stride + 1) + 1;
            for (int x = 0; x < *w; ++x) {
                const uint8_t* p = row + (size_t)x * ch;
                uint8_t v;
                if (ctype == 0 || ctype == 4) v = p[0];
                else if (ctype == 3) v = luma(palette[p[0]][0], palette[p[0]][1], palette[p[0]][2]);
                else v = luma(p[0], p[1], p[2]);
                out[(size_t)y * *w + x] = v;
            }
        }
    free(raw);
    return out;
}

uint8_t* sgm_load_gray(const char* path, int* w, int* h)
{
    size_t n = 0;
    uint8_t* b = read_file(path, &n);
    if (!b) return NULL;
    uint8_t* out = NULL;
    if (n > 2 && b[0] == 'P' && (b[1] == '5' || b[1] == '6')) out = load_pnm(b, n, w, h);
    else out = load_png(b, n, w, h);
    free(b);
    return out;
}

/* ---------------------------------------------------------------- writers */

static void put_chunk(FILE* f, const char* type, const uint8_t* data, uint32_t len)
{
    uint8_t hdr[8] = {(uint8_t)(len >> 24), (uint8_t)(len >> 16), (uint8_t)(len >> 8), (uint8_t)len,
                      (uint8_t)type[0], (uint8_t)type[1], (uint8_t)type[2], (uint8_t)type[3]};
    uLong crc = crc32(0L, hdr + 4, 4);
    if (len) crc = crc32(crc, data, len);
    const uint8_t tail[4] = {(uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc};
    fwrite(hdr, 1, 8, f);
    if (len) fwrite(data, 1, len, f);
    fwrite(tail, 1, 4, f);
}

int sgm_write_png_gray(const char* path, const uint8_t* data, int w, int h)
{
    const size_t stride = (size_t)w + 1;
    uint8_t* raw = (uint8_t*)malloc(stride * (size_t)h);
    uLongf clen = compressBound((uLong)(stride * (size_t)h));
    uint8_t* comp = (uint8_t*)malloc(clen);
    if (!raw || !comp) { free(raw); free(comp); return -1; }
    for (int y = 0; y < h; ++y) {
        raw[(size_t)y * stride] = 0;                     /* filter type 0 */
        memcpy(raw + (size_t)y * stride + 1, data + (size_t)y * w, (size_t)w);
    }
    int rc = -1;
    FILE* f = NULL;
    if (compress2(comp, &clen, raw, (uLong)(stride * (size_t)h), 6) == Z_OK && (f = fopen(path, "wb"))) {
        static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
        const uint8_t ihdr[13] = {(uint8_t)(w >> 24), (uint8_t)(w >> 16), (uint8_t)(w >> 8), (uint8_t)w,
                                  (uint8_t)(h >> 24), (uint8_t)(h >> 16), (uint8_t)(h >> 8), (uint8_t)h, 8, 0, 0, 0, 0};
        fwrite(sig, 1, 8, f);
        put_chunk(f, "IHDR", ihdr, 13);
        put_chunk(f, "IDAT", comp, (uint32_t)clen);
        put_chunk(f, "IEND", NULL, 0);
        rc = fclose(f) == 0 ? 0 : -1;
    }
    free(raw); free(comp);
    return rc;
}

int sgm_write_pgm(const char* path, const uint8_t* data, int w, int h)
{
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    fprintf(f, "P5\n%d %d\n255\n", w, h);
    fwrite(data, 1, (size_t)w * h, f);
    return fclose(f) == 0 ? 0 : -1;
}
