/* p8front/p8f_xml.c -- HOST FRONT END of the paq8 stage (product code; tables are recorded through p8f_emit.h, the device learns).
 *
 * Host front end for paq8's XMLModel (reference src/models/paq8.cpp:7823-8096): a tag-level state machine (tag names,
 * attributes, content, CDATA, comments) over a cache of the last 32 tags, content-type detection (dates, times, URLs,
 * numbers, coordinates, temperatures, ISBN), indentation tracking; four contexts per byte into one ContextMap, and the
 * Stats.XML byte other parts of paq8 read. enwik-type input is XML, so this one matters for the headline workload.
 * Parity: tests/test_p8stage_host.py (stage vs columns 434..2024 of reference traces). */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct CM1 CM1;
CM1* p8f_cm_new(uint64_t size_bytes, int count);
int p8f_cm_step(CM1* c, int y1, int bp, int c0, int c1, const uint64_t* ctx, int nset, int16_t* out, int* nout);
uint64_t p8f_hash2(uint64_t a, uint64_t b);
uint64_t p8f_hash3(uint64_t a, uint64_t b, uint64_t c);
uint64_t p8f_hash4(uint64_t a, uint64_t b, uint64_t c, uint64_t d);
uint64_t p8f_hash5(uint64_t a, uint64_t b, uint64_t c, uint64_t d, uint64_t e);

enum { kCache = 32 };
enum { F_Text = 1, F_Number = 2, F_Date = 4, F_Time = 8, F_URL = 16, F_Link = 32, F_Coord = 64, F_Temp = 128, F_ISBN = 256 };
enum { S_None, S_TagName, S_Tag, S_AttrName, S_AttrValue, S_Content, S_CDATA, S_Comment };

typedef struct { uint32_t Name, Value, Length; } Attr;
typedef struct { uint32_t Data, Length, Type; } Content;
typedef struct {
  uint32_t Name, Length;
  int Level;
  uint8_t EndTag, Empty;
  Content content;
  Attr items[4];
  uint32_t attrIndex;
} Tag;
typedef struct {
  CM1* cm;
  Tag tags[kCache];
  uint32_t index;
  uint32_t stateBH[8];
  int state, pstate;
  uint32_t c8, wsRun, pWsRun, indentTab, indentStep, lineEnding;
} Xml;

Xml* p8f_xml_new(int level) {
  Xml* x = (Xml*)calloc(1, sizeof *x);
  x->cm = p8f_cm_new((0x10000ull << level) / 4, 4);
  x->indentStep = 2;
  x->lineEnding = 2;
  return x;
}

#define RB(i) ((uint32_t)hist[((uint32_t)pos - (uint32_t)(i)) & bmask])
static int digit(uint32_t v) { return v >= 0x30 && v <= 0x39; }
static void detect_content(Content* ct, uint32_t c4, uint32_t c8, uint8_t B, const uint8_t* hist, uint32_t bmask, int pos) {
  if ((c4 & 0xF0F0F0F0) == 0x30303030) {
    int i = 0;
    while (i < 4 && digit((c4 >> (8 * i)) & 0xFF)) i++;
    if (i == 4 && (((c8 & 0xFDF0F0FD) == 0x2D30302D && digit(RB(9))) || ((c8 & 0xF0FDF0FD) == 0x302D302D))) ct->Type |= F_Date;
  } else if (((c8 & 0xF0F0FDF0) == 0x30302D30 || (c8 & 0xF0F0F0FD) == 0x3030302D) && digit(RB(9))) {
    int i = 2;
    while (i < 4 && digit((c8 >> (8 * i)) & 0xFF)) i++;
    if (i == 4 && (c4 & 0xF0FDF0F0) == 0x302D3030) ct->Type |= F_Date;
  }
  if ((c4 & 0xF0FFF0F0) == 0x303A3030 && digit(RB(5)) && (!digit(RB(6)) || ((c8 & 0xF0F0FF00) == 0x30303A00 && !digit(RB(9)))))
    ct->Type |= F_Time;
  if (ct->Length >= 8 && (c8 & 0x80808080) == 0 && (c4 & 0x80808080) == 0) ct->Type |= F_Text;
  if ((c8 & 0xF0F0FF) == 0x3030C2 && (c4 & 0xFFF0F0FF) == 0xB0303027) {
    int i = 2;
    while (i < 7 && digit(RB(i))) i += (i & 1) * 2 + 1;
    if (i == 10) ct->Type |= F_Coord;
  }
  if ((c4 & 0xFFFFFA) == 0xC2B042 && B != 0x47 && (digit(c4 >> 24) || ((c4 >> 24) == 0x20 && digit(RB(5))))) ct->Type |= F_Temp;
  if (digit(B)) ct->Type |= F_Number;
  if (c4 == 0x4953424E && RB(5) == 0x20) ct->Type |= F_ISBN;
}

int p8f_xml_step(Xml* x, int y, int bpos, int c0, uint32_t c4, const uint8_t* hist, uint32_t bmask, int pos, int16_t* out,
                    uint32_t* xml_out) {
  uint64_t cx[4];
  int nset = 0;
  if (bpos == 0) {
    const uint8_t B = (uint8_t)c4;
    Tag* pTag = &x->tags[(x->index - 1) & (kCache - 1)];
    Tag* tag = &x->tags[x->index & (kCache - 1)];
    Attr* attr = &tag->items[tag->attrIndex & 3];
    Content* ct = &tag->content;
    x->pstate = x->state;
    x->c8 = (x->c8 << 8) | RB(5);
    const uint32_t c8 = x->c8;
    if ((B == 0x09 || B == 0x20) && (B == (uint8_t)(c4 >> 8) || !x->wsRun)) {
      x->wsRun++;
      x->indentTab = (B == 0x09);
    } else {
      if ((x->state == S_None || (x->state == S_Content && ct->Length <= x->lineEnding + x->wsRun)) && x->wsRun > 1 + x->indentTab &&
          x->wsRun != x->pWsRun) {
        x->indentStep = (uint32_t)abs((int)(x->wsRun - x->pWsRun));
        x->pWsRun = x->wsRun;
      }
      x->wsRun = 0;
    }
    if (B == 0x0A) x->lineEnding = 1 + ((uint8_t)(c4 >> 8) == 0x0D);
    const int pS = x->pstate;
    switch (x->state) {
      case S_None:
        if (B == 0x3C) {
          x->state = S_TagName;
          memset(tag, 0, sizeof *tag);
          tag->Level = (pTag->EndTag || pTag->Empty) ? pTag->Level : pTag->Level + 1;
        }
        if (tag->Level > 1) detect_content(ct, c4, c8, B, hist, bmask, pos);
        cx[nset++] = p8f_hash3((uint64_t)pS, (uint64_t)x->state,
                                  (uint64_t)(uint32_t)((uint32_t)(pTag->Level + 1) * x->indentStep - x->wsRun));
        break;
      case S_TagName: {
        if (tag->Length > 0 && (B == 0x09 || B == 0x0A || B == 0x0D || B == 0x20)) x->state = S_Tag;
        else if ((B == 0x3A || (B >= 'A' && B <= 'Z') || B == 0x5F || (B >= 'a' && B <= 'z')) ||
                 (tag->Length > 0 && (B == 0x2D || B == 0x2E || (B >= '0' && B <= '9')))) {
          tag->Length++;
          tag->Name = tag->Name * 263 * 32 + (B & 0xDF);
        } else if (B == 0x3E) {
          if (tag->EndTag) { x->state = S_None; x->index++; }
          else x->state = S_Content;
        } else if (B != 0x21 && B != 0x2D && B != 0x2F && B != 0x5B) { x->state = S_None; x->index++; }
        else if (tag->Length == 0) {
          if (B == 0x2F) { tag->EndTag = 1; tag->Level = tag->Level - 1 > 0 ? tag->Level - 1 : 0; }
          else if (c4 == 0x3C212D2D) { x->state = S_Comment; tag->Level = tag->Level - 1 > 0 ? tag->Level - 1 : 0; }
        }
        if (tag->Length == 1 && (c4 & 0xFFFF00) == 0x3C2100) { memset(tag, 0, sizeof *tag); x->state = S_None; }
        else if (tag->Length == 5 && c8 == 0x215B4344 && c4 == 0x4154415B) {
          x->state = S_CDATA;
          tag->Level = tag->Level - 1 > 0 ? tag->Level - 1 : 0;
        }
        int i = 1;
        do {
          pTag = &x->tags[(x->index - (uint32_t)i) & (kCache - 1)];
          i += 1 + (pTag->EndTag && x->tags[(x->index - (uint32_t)i - 1) & (kCache - 1)].Name == pTag->Name);
        } while (i < kCache && (pTag->EndTag || pTag->Empty));
        cx[nset++] = p8f_hash5((uint64_t)(pS * 8 + x->state), tag->Name, (uint64_t)(int64_t)tag->Level, pTag->Name,
                                  (uint64_t)(pTag->Level != tag->Level));
        break;
      }
      case S_Tag:
        if (B == 0x2F) tag->Empty = 1;
        else if (B == 0x3E) {
          if (tag->Empty) { x->state = S_None; x->index++; }
          else x->state = S_Content;
        } else if (B != 0x09 && B != 0x0A && B != 0x0D && B != 0x20) { x->state = S_AttrName; attr->Name = B & 0xDF; }
        cx[nset++] = p8f_hash5((uint64_t)pS, (uint64_t)x->state, tag->Name, B, tag->attrIndex);
        break;
      case S_AttrName:
        if ((c4 & 0xFFF0) == 0x3D20 && (B == 0x22 || B == 0x27)) {
          x->state = S_AttrValue;
          if ((c8 & 0xDFDF) == 0x4852 && (c4 & 0xDFDF0000) == 0x45460000) ct->Type |= F_Link;
        } else if (B != 0x22 && B != 0x27 && B != 0x3D) attr->Name = attr->Name * 263 * 32 + (B & 0xDF);
        cx[nset++] = p8f_hash5((uint64_t)(pS * 8 + x->state), attr->Name, tag->attrIndex, tag->Name, ct->Type);
        break;
      case S_AttrValue:
        if (B == 0x22 || B == 0x27) { tag->attrIndex++; x->state = S_Tag; }
        else {
          attr->Value = attr->Value * 263 * 32 + (B & 0xDF);
          attr->Length++;
          if ((c8 & 0xDFDFDFDF) == 0x48545450 && ((c4 >> 8) == 0x3A2F2F || c4 == 0x733A2F2F)) ct->Type |= F_URL;
        }
        cx[nset++] = p8f_hash4((uint64_t)pS, (uint64_t)x->state, attr->Name, ct->Type);
        break;
      case S_Content:
        if (B == 0x3C) {
          x->state = S_TagName;
          x->index++;
          Tag* nt = &x->tags[x->index & (kCache - 1)];
          memset(nt, 0, sizeof *nt);
          nt->Level = tag->Level + 1;
        } else {
          ct->Length++;
          ct->Data = ct->Data * 997 * 16 + (B & 0xDF);
          detect_content(ct, c4, c8, B, hist, bmask, pos);
        }
        cx[nset++] = p8f_hash4((uint64_t)pS, (uint64_t)x->state, tag->Name, c4 & 0xC0FF);
        break;
      case S_CDATA:
        if ((c4 & 0xFFFFFF) == 0x5D5D3E) { x->state = S_None; x->index++; }
        cx[nset++] = p8f_hash2((uint64_t)pS, (uint64_t)x->state);
        break;
      default:  /* S_Comment */
        if ((c4 & 0xFFFFFF) == 0x2D2D3E) { x->state = S_None; x->index++; }
        cx[nset++] = p8f_hash2((uint64_t)pS, (uint64_t)x->state);
        break;
    }
    x->stateBH[x->state] = (x->stateBH[x->state] << 8) | B;
    pTag = &x->tags[(x->index - 1) & (kCache - 1)];
    uint64_t i = 64;
    ++i; cx[nset++] = p8f_hash5(i, (uint64_t)x->state, (uint64_t)(int64_t)tag->Level, (uint64_t)(pS * 2 + tag->EndTag), tag->Name);
    ++i; cx[nset++] = p8f_hash5(i, pTag->Name, (uint64_t)(x->state * 2 + pTag->EndTag), pTag->content.Type, tag->content.Type);
    ++i; cx[nset++] = p8f_hash5(i, (uint64_t)(x->state * 2 + tag->EndTag), tag->Name, tag->content.Type, c4 & 0xE0FF);
  }
  int nout = 0;
  p8f_cm_step(x->cm, y, bpos, c0, (int)RB(1), cx, nset, out, &nout);
  const uint32_t bh = x->stateBH[x->state];
  const uint8_t s = (uint8_t)(((bh >> (28 - bpos)) & 0x08) | ((bh >> (21 - bpos)) & 0x04) | ((bh >> (14 - bpos)) & 0x02) |
                              ((bh >> (7 - bpos)) & 0x01) | (bpos << 4));
  *xml_out = ((uint32_t)s << 3) | (uint32_t)x->state;
  return nout;
}
