/* p8front/p8f_exe.c -- HOST FRONT END of the paq8 stage (product code; tables are recorded through p8f_emit.h, the device learns).
 *
 * Host front end for paq8's exeModel (reference src/models/paq8.cpp:6560-7546): an x86/x64 instruction-boundary decoder
 * (prefixes, REX, 1/2/3-byte opcodes, ModRM/SIB, immediates and displacements; opcode tables in p8f_tables.h,
 * dumped from the reference build) whose parser state, a cache of the last 32 quantised instructions and sparse
 * byte contexts feed a 20-context ContextMap2 and six mixer weight-set selectors. contextModel2 runs it with
 * Forced = true on every input, text included. Parity: tests/test_p8stage_host.py (stage vs columns 434..2024 of reference traces). */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "p8f_tables.h"

typedef struct CM2 CM2;
CM2* p8f_cm2_new(uint64_t size_bytes, uint32_t count);
int p8f_cm2_step(CM2* c, int y_prev, int bpos, const uint64_t* ctx, int nset, int16_t* out, int* nout);
uint32_t p8f_finalize64(uint64_t h, int bits);

#define PHI64 0x9E3779B97F4A7C15ull
static const uint64_t MUL[8] = {PHI64, 0x993DDEFFB1462949ull, 0xE9C91DC159AB0D2Dull, 0x83D6A14F1B0CED73ull,
                                0xA14F1B0CED5A841Full, 0xC0E51314A614F4EFull, 0xDA9CC2600AE45A27ull, 0x826797AA04A65737ull};
static uint64_t hn(int n, const uint64_t* x) { uint64_t h = 0; for (int i = 0; i < n; ++i) h += (x[i] + 1) * MUL[i]; return h; }
#define U(x) ((uint64_t)(x))
#define H2(a, b) hn(2, (const uint64_t[]){U(a), U(b)})
#define H3(a, b, c) hn(3, (const uint64_t[]){U(a), U(b), U(c)})
#define H4(a, b, c, d) hn(4, (const uint64_t[]){U(a), U(b), U(c), U(d)})
#define H5(a, b, c, d, e) hn(5, (const uint64_t[]){U(a), U(b), U(c), U(d), U(e)})
#define H6(a, b, c, d, e, f) hn(6, (const uint64_t[]){U(a), U(b), U(c), U(d), U(e), U(f)})

/* InstructionFormat :6584-6606 */
enum { fNM = 0, fAM = 1, fMR = 2, fMEXTRA = 3, fMODE = 3, fNI = 0, fBI = 4, fWI = 8, fDI = 0xc, fTYPE = 0xc, fAD = 0, fDA = 4, fBR = 8,
       fDR = 0xc, fERR = 0xf };
enum { St_Start, St_PrefOpSize, St_PrefMulti, St_ParseFlags, St_ExtraFlags, St_ReadModRM, St_OP3_38, St_OP3_3A, St_ReadSIB, St_Read8,
       St_Read16, St_Read32, St_Read8_ModRM, St_Read16_f, St_Read32_ModRM, St_Error };
enum { kCache = 32, CodeShift = 3, CategoryShift = 5, MinRequired = 8 };
#define CodeMask (0xFFu << CodeShift)
#define ClearCodeMask (0xFFFFFFFFu ^ CodeMask)
#define PrefixMask ((1u << CodeShift) - 1)
#define OperandSizeOverride (0x01u << (8 + CodeShift))
#define MultiByteOpcode (0x02u << (8 + CodeShift))
#define PrefixREX (0x04u << (8 + CodeShift))
#define Prefix38 (0x08u << (8 + CodeShift))
#define Prefix3A (0x10u << (8 + CodeShift))
#define HasExtraFlags (0x20u << (8 + CodeShift))
#define HasModRM (0x40u << (8 + CodeShift))
#define ModRMShift (7 + 8 + CodeShift)
#define SIBScaleShift (ModRMShift + 8 - 6)
#define RegDWordDisplacement (0x01u << (8 + SIBScaleShift))
#define AddressMode (0x02u << (8 + SIBScaleShift))
#define TypeShift (2 + 8 + SIBScaleShift)
#define CategoryMask ((1u << CategoryShift) - 1)
#define ModRM_mod 0xC0
#define ModRM_reg 0x38
#define ModRM_rm 0x07
#define SIB_scale 0xC0
#define SIB_base 0x07
#define REX_w 0x08

typedef struct {
  uint32_t Data;
  uint8_t Prefix, Code, ModRM, SIB, REX, Flags, BytesRead, Size, Category;
  uint8_t MustCheckREX, Decoding, o16, imm8;
} Instr;
typedef struct {
  CM2* cm;
  uint32_t cacheOp[kCache], cacheIndex;
  uint32_t stateBH[256];
  int pState, State;
  Instr Op;
  uint32_t TotalOps, OpMask, OpCategMask, Context, BrkPoint;
  uint32_t BrkCtx;  /* the reference keeps only the low 32 bits of these hashes (static U32, :7280) */
  int Valid;
} Exe;

Exe* p8f_exe_new(int level) {
  Exe* e = (Exe*)calloc(1, sizeof *e);
  e->cm = p8f_cm2_new((0x10000ull << level) * 2, 20);
  return e;
}
static int invalid_x64(uint8_t op) { for (int i = 0; i < 19; i++) if (op == P8_EXE_INVALID_X64[i]) return 1; return 0; }
static int valid_x64_prefix(uint8_t p) {
  for (int i = 0; i < 8; i++) if (p == P8_EXE_X64_PREFIXES[i]) return 1;
  return (p >= 0x40 && p <= 0x4F) || (p >= 0x64 && p <= 0x67);
}
static void process_mode(Instr* Op, int* State) {  /* :7155-7201 */
  if ((Op->Flags & fMODE) == fAM) {
    Op->Data |= AddressMode;
    Op->BytesRead = 0;
    switch (Op->Flags & fTYPE) {
      case fDR: Op->Data |= (2u << TypeShift);  /* falls through, as in the reference */
      case fDA: Op->Data |= (1u << TypeShift);  /* falls through */
      case fAD: *State = St_Read32; break;
      case fBR: Op->Data |= (2u << TypeShift); *State = St_Read8;
    }
  } else {
    switch (Op->Flags & fTYPE) {
      case fBI: *State = St_Read8; break;
      case fWI: *State = St_Read16; Op->Data |= (1u << TypeShift); Op->BytesRead = 0; break;
      case fDI:
        Op->imm8 = ((Op->REX & REX_w) > 0 && (Op->Code & 0xF8) == 0xB8);
        if (!Op->o16 || Op->imm8) { *State = St_Read32; Op->Data |= (2u << TypeShift); }
        else { *State = St_Read16; Op->Data |= (3u << TypeShift); }
        Op->BytesRead = 0;
        break;
      default: *State = St_Start;
    }
  }
}
static void process_flags2(Instr* Op, int* State) {
  if ((Op->Flags & fMODE) == fMR && *State != St_ExtraFlags) { *State = St_ReadModRM; return; }
  process_mode(Op, State);
}
static void process_flags(Instr* Op, int* State) {
  if (Op->Code == 0x9a || Op->Code == 0xea || Op->Code == 0xc8) { Op->BytesRead = 0; *State = St_Read16_f; return; }
  process_flags2(Op, State);
}
static void check_flags(Instr* Op, int* State) {
  if (Op->Flags == fMEXTRA) *State = St_ExtraFlags;
  else if (Op->Flags == fERR) { memset(Op, 0, sizeof *Op); *State = St_Error; }
  else process_flags(Op, State);
}
static void read_flags(Instr* Op, int* State) {
  Op->Flags = P8_EXE_TABLE1[Op->Code];
  Op->Category = P8_EXE_TYPEOP1[Op->Code];
  check_flags(Op, State);
}
static void process_modrm(Instr* Op, int* State) {
  if ((Op->ModRM & ModRM_mod) == 0x40) *State = St_Read8_ModRM;
  else if ((Op->ModRM & ModRM_mod) == 0x80 || (Op->ModRM & (ModRM_mod | ModRM_rm)) == 0x05 || (Op->ModRM < 0x40 && (Op->SIB & SIB_base) == 0x05)) {
    *State = St_Read32_ModRM;
    Op->BytesRead = 0;
  } else process_mode(Op, State);
}
static void apply_code(Instr* Op, uint32_t flag) { Op->Data &= ClearCodeMask; Op->Data |= ((uint32_t)Op->Code << CodeShift) | flag; }
#define OPN(n) (e->cacheOp[(e->cacheIndex - (n)) & (kCache - 1)])
#define RB(i) ((int)hist[((uint32_t)pos - (uint32_t)(i)) & bmask])
static int pref(const uint8_t* hist, uint32_t bmask, int pos, int i) { return (RB(i) == 0x0f) + 2 * (RB(i) == 0x66) + 3 * (RB(i) == 0x67); }
static uint32_t execxt(const uint8_t* hist, uint32_t bmask, int pos, int i, int x) {  /* :7255-7263 */
  int prefix = 0, opcode = 0, modrm = 0, sib = 0;
  if (i) prefix += 4 * pref(hist, bmask, pos, i--);
  if (i) prefix += pref(hist, bmask, pos, i--);
  if (i) opcode += RB(i--);
  if (i) modrm += RB(i--) & (ModRM_mod | ModRM_rm);
  if (i && ((modrm & ModRM_rm) == 4) && (modrm < ModRM_mod)) sib = RB(i) & SIB_scale;
  return (uint32_t)(prefix | opcode << 4 | modrm << 12 | x << 20 | sib << (28 - 6));
}

int p8f_exe_step(Exe* e, int y, int bpos, int c0, uint32_t c4, int blpos, const uint8_t* hist, uint32_t bmask, int pos,
                    int16_t* out, int* sets, uint32_t* x86_out) {
  uint64_t cx[20];
  int nset = 0;
  Instr* Op = &e->Op;
  if (!bpos) {
    e->pState = e->State;
    const uint8_t B = (uint8_t)c4;
    Op->Size++;
    switch (e->State) {
      case St_Start: case St_Error: {
        int Skip = 0, brk = 0;
        if (Op->MustCheckREX) {
          Op->MustCheckREX = 0;
          if (!invalid_x64(B) && !valid_x64_prefix(B)) {
            Op->REX = Op->Code;
            Op->Code = B;
            Op->Data = PrefixREX | ((uint32_t)Op->Code << CodeShift) | (Op->Data & PrefixMask);
            Skip = 1;
          }
        }
        Op->ModRM = Op->SIB = Op->REX = Op->Flags = Op->BytesRead = 0;
        if (!Skip) {
          Op->Code = B;
          Op->MustCheckREX = ((Op->Code & 0xF0) == 0x40) && (!(Op->Decoding && ((Op->Data & PrefixMask) == 1)));
          Op->Prefix = (uint8_t)((Op->Code == 0x26 || Op->Code == 0x2E || Op->Code == 0x36 || Op->Code == 0x3E) + (Op->Code == 0x64) * 2 +
                                 (Op->Code == 0x65) * 3 + (Op->Code == 0x67) * 4 + (Op->Code == 0x9B) * 5 + (Op->Code == 0xF0) * 6 +
                                 (Op->Code == 0xF2 || Op->Code == 0xF3) * 7);
          if (!Op->Decoding) {
            e->TotalOps += (uint32_t)((Op->Data != 0) - (e->cacheIndex && e->cacheOp[e->cacheIndex & (kCache - 1)] != 0));
            e->OpMask = (e->OpMask << 1) | (e->State != St_Error);
            e->OpCategMask = (e->OpCategMask << CategoryShift) | Op->Category;
            Op->Size = 0;
            e->cacheOp[e->cacheIndex & (kCache - 1)] = Op->Data;
            e->cacheIndex++;
            if (!Op->Prefix) Op->Data = (uint32_t)Op->Code << CodeShift;
            else {
              Op->Data = Op->Prefix;
              Op->Category = P8_EXE_TYPEOP1[Op->Code];
              Op->Decoding = 1;
              e->BrkPoint = 0;
              e->BrkCtx = (uint32_t)H3(1 + 0, Op->Prefix, e->OpCategMask & CategoryMask);
              brk = 1;
            }
          } else {
            if (!Op->Prefix) { Op->Data |= ((uint32_t)Op->Code << CodeShift); Op->Decoding = 0; }
            else {
              Op->Data = Op->Prefix;
              Op->Category = P8_EXE_TYPEOP1[Op->Code];
              e->BrkPoint = 1;
              e->BrkCtx = (uint32_t)H3(1 + 1, Op->Prefix, e->OpCategMask & CategoryMask);
              brk = 1;
            }
          }
        }
        if (brk) break;
        if ((Op->o16 = (Op->Code == 0x66))) e->State = St_PrefOpSize;
        else if (Op->Code == 0x0f) e->State = St_PrefMulti;
        else read_flags(Op, &e->State);
        e->BrkPoint = 2;
        e->BrkCtx = (uint32_t)H5(1 + 2, e->State, Op->Code, e->OpCategMask & CategoryMask,
                       OPN(1) & ((uint32_t)(ModRM_mod | ModRM_reg | ModRM_rm) << ModRMShift));
        break;
      }
      case St_PrefOpSize:
        Op->Code = B;
        apply_code(Op, OperandSizeOverride);
        read_flags(Op, &e->State);
        e->BrkPoint = 3; e->BrkCtx = (uint32_t)H2(1 + 3, e->State);
        break;
      case St_PrefMulti:
        Op->Code = B;
        Op->Data |= MultiByteOpcode;
        if (Op->Code == 0x38) e->State = St_OP3_38;
        else if (Op->Code == 0x3A) e->State = St_OP3_3A;
        else {
          apply_code(Op, 0);
          Op->Flags = P8_EXE_TABLE2[Op->Code];
          Op->Category = P8_EXE_TYPEOP2[Op->Code];
          check_flags(Op, &e->State);
        }
        e->BrkPoint = 4; e->BrkCtx = (uint32_t)H2(1 + 4, e->State);
        break;
      case St_ParseFlags:
        process_flags(Op, &e->State);
        e->BrkPoint = 5; e->BrkCtx = (uint32_t)H2(1 + 5, e->State);
        break;
      case St_ExtraFlags: case St_ReadModRM:
        Op->ModRM = B;
        Op->Data |= ((uint32_t)Op->ModRM << ModRMShift) | HasModRM;
        Op->SIB = 0;
        if (Op->Flags == fMEXTRA) {
          Op->Data |= HasExtraFlags;
          const int i = ((Op->ModRM >> 3) & 0x07) | ((Op->Code & 0x01) << 3) | ((Op->Code & 0x08) << 1);
          Op->Flags = P8_EXE_TABLEX[i];
          Op->Category = P8_EXE_TYPEOPX[i];
          if (Op->Flags == fERR) {
            memset(Op, 0, sizeof *Op);
            e->State = St_Error;
            e->BrkPoint = 6; e->BrkCtx = (uint32_t)H2(1 + 6, e->State);
            break;
          }
          process_flags(Op, &e->State);
          e->BrkPoint = 7; e->BrkCtx = (uint32_t)H2(1 + 7, e->State);
          break;
        }
        if ((Op->ModRM & ModRM_rm) == 4 && Op->ModRM < ModRM_mod) {
          e->State = St_ReadSIB;
          e->BrkPoint = 8; e->BrkCtx = (uint32_t)H2(1 + 8, e->State);
          break;
        }
        process_modrm(Op, &e->State);
        e->BrkPoint = 9; e->BrkCtx = (uint32_t)H3(1 + 9, e->State, Op->Code);
        break;
      case St_OP3_38: case St_OP3_3A:
        Op->Code = B;
        apply_code(Op, Prefix38 << (e->State - St_OP3_38));
        if (e->State == St_OP3_38) { Op->Flags = P8_EXE_TABLE3_38[Op->Code]; Op->Category = P8_EXE_TYPEOP3_38[Op->Code]; }
        else { Op->Flags = P8_EXE_TABLE3_3A[Op->Code]; Op->Category = P8_EXE_TYPEOP3_3A[Op->Code]; }
        check_flags(Op, &e->State);
        e->BrkPoint = 10; e->BrkCtx = (uint32_t)H2(1 + 10, e->State);
        break;
      case St_ReadSIB:
        Op->SIB = B;
        Op->Data |= ((uint32_t)(Op->SIB & SIB_scale) << SIBScaleShift);
        process_modrm(Op, &e->State);
        e->BrkPoint = 11; e->BrkCtx = (uint32_t)H3(1 + 11, e->State, Op->SIB & SIB_scale);
        break;
      case St_Read8: case St_Read16: case St_Read32:
        if (++Op->BytesRead >= ((2 * (e->State - St_Read8)) << Op->imm8)) { Op->BytesRead = 0; Op->imm8 = 0; e->State = St_Start; }
        e->BrkPoint = 12;
        e->BrkCtx = (uint32_t)H5(1 + 12, e->State, Op->Flags & fMODE, Op->BytesRead,
                       ((Op->BytesRead > 1) ? (RB(Op->BytesRead) << 8) : 0) | ((Op->BytesRead) ? B : 0));
        break;
      case St_Read8_ModRM:
        process_mode(Op, &e->State);
        e->BrkPoint = 13; e->BrkCtx = (uint32_t)H2(1 + 13, e->State);
        break;
      case St_Read16_f:
        if (++Op->BytesRead == 2) { Op->BytesRead = 0; process_flags2(Op, &e->State); }
        e->BrkPoint = 14; e->BrkCtx = (uint32_t)H2(1 + 14, e->State);
        break;
      case St_Read32_ModRM:
        Op->Data |= RegDWordDisplacement;
        if (++Op->BytesRead == 4) { Op->BytesRead = 0; process_mode(Op, &e->State); }
        e->BrkPoint = 15; e->BrkCtx = (uint32_t)H2(1 + 15, e->State);
        break;
    }
    e->Valid = (e->TotalOps > 2 * MinRequired) && ((e->OpMask & ((1u << MinRequired) - 1)) == ((1u << MinRequired) - 1));
    e->Context = (uint32_t)e->State + 16u * Op->BytesRead + 16u * (Op->REX & REX_w);
    e->stateBH[e->Context] = (e->stateBH[e->Context] << 8) | B;
    {  /* Valid || Forced: always, the caller forces it */
      int mask = 0, count0 = 0, i = 0;
      for (int j = 0; i < 10; ++i) {
        if (i > 1) { mask = mask * 2 + (RB(i - 1) == 0); count0 += mask & 1; }
        j = (i < 4) ? i + 1 : 5 + (i - 4) * (2 + (i > 6));
        cx[nset++] = H4(i, execxt(hist, bmask, pos, j, RB(1) * (j > 6)), ((1 << 10) | mask) * (count0 * 10 / 2 >= i),
                        (0x08 | (blpos & 0x07)) * (i < 4));
      }
      cx[nset++] = e->BrkCtx;
      const int St = e->State;
      uint32_t m = PrefixMask | (0xF8u << CodeShift) | MultiByteOpcode | Prefix38 | Prefix3A;
      ++i; cx[nset++] = H6(i, OPN(1) & (m | RegDWordDisplacement | AddressMode), St + 16 * Op->BytesRead, Op->Data & m, Op->REX, Op->Category);
      m = 0x04u | (0xFEu << CodeShift) | MultiByteOpcode | Prefix38 | Prefix3A | ((uint32_t)(ModRM_mod | ModRM_reg) << ModRMShift);
      ++i; cx[nset++] = H6(i, OPN(1) & m, OPN(2) & m, OPN(3) & m, e->Context + 256u * ((Op->ModRM & ModRM_mod) == ModRM_mod),
                           Op->Data & ((m | PrefixREX) ^ ((uint32_t)ModRM_mod << ModRMShift)));
      m = 0x04u | CodeMask;
      ++i; cx[nset++] = H6(i, OPN(1) & m, OPN(2) & m, OPN(3) & m, OPN(4) & m, (Op->Data & m) | ((uint32_t)St << 11) | ((uint32_t)Op->BytesRead << 15));
      m = 0x04u | (0xFCu << CodeShift) | MultiByteOpcode | Prefix38 | Prefix3A;
      ++i; cx[nset++] = H6(i, St + 16 * Op->BytesRead, Op->Data & m, Op->Category * 8 + (e->OpMask & 0x07), Op->Flags,
                           ((Op->SIB & SIB_base) == 5) * 4 + ((Op->ModRM & ModRM_reg) == ModRM_reg) * 2 + ((Op->ModRM & ModRM_mod) == 0));
      m = PrefixMask | CodeMask | OperandSizeOverride | MultiByteOpcode | PrefixREX | Prefix38 | Prefix3A | HasExtraFlags | HasModRM |
          ((uint32_t)(ModRM_mod | ModRM_rm) << ModRMShift);
      ++i; cx[nset++] = H4(i, Op->Data & m, St + 16 * Op->BytesRead, Op->Flags);
      m = PrefixMask | CodeMask | OperandSizeOverride | MultiByteOpcode | Prefix38 | Prefix3A | HasExtraFlags | HasModRM;
      ++i; cx[nset++] = H5(i, OPN(1) & m, St, Op->BytesRead * 2 + ((Op->REX & REX_w) > 0), Op->Data & ((uint16_t)(m ^ OperandSizeOverride)));
      m = 0x04u | (0xFEu << CodeShift) | MultiByteOpcode | Prefix38 | Prefix3A | ((uint32_t)ModRM_reg << ModRMShift);
      ++i; cx[nset++] = H5(i, OPN(1) & m, OPN(2) & m, St + 16 * Op->BytesRead, Op->Data & (m | PrefixMask | CodeMask));
      ++i; cx[nset++] = H2(i, St + 16 * Op->BytesRead);
      ++i; cx[nset++] = H4(i, (0x100 | B) * (Op->BytesRead > 0), St + 16 * e->pState + 256 * Op->BytesRead,
                           ((Op->Flags & fMODE) == fAM) * 16 + (Op->REX & REX_w) + (Op->o16) * 4 + ((Op->Code & 0xFE) == 0xE8) * 2 +
                               ((Op->Data & MultiByteOpcode) != 0 && (Op->Code & 0xF0) == 0x80));
    }
  }
  int nout = 0;
  p8f_cm2_step(e->cm, y, bpos, cx, nset, out, &nout);
  const uint32_t bh = e->stateBH[e->Context];
  const uint8_t s = (uint8_t)(((bh >> (28 - bpos)) & 0x08) | ((bh >> (21 - bpos)) & 0x04) | ((bh >> (14 - bpos)) & 0x02) | ((bh >> (7 - bpos)) & 0x01) |
                              ((Op->Category == P8_EXE_OP_GEN_BRANCH) << 4) | (((c0 & ((1 << bpos) - 1)) == 0) << 5));
  const int St = e->State;
  sets[0] = (int)(e->Context * 4 + (s >> 4));
  sets[1] = 1024 + St * 64 + bpos * 8 + (Op->BytesRead > 0) * 4 + (s >> 4);
  sets[2] = 2048 + (int)((e->BrkCtx & 0x1FF) | ((uint32_t)(s & 0x20) << 4));
  sets[3] = 3072 + (int)p8f_finalize64(H3(Op->Code, St, OPN(1) & CodeMask), 13);
  sets[4] = 3072 + 8192 + (int)p8f_finalize64(H4(St, bpos, Op->Code, Op->BytesRead), 13);
  sets[5] = 3072 + 2 * 8192 + (int)p8f_finalize64(H4(St, (bpos << 2) | (c0 & 3), e->OpCategMask & CategoryMask,
                                                      ((Op->Category == P8_EXE_OP_GEN_BRANCH) << 2) | (((Op->Flags & fMODE) == fAM) << 1) | (Op->BytesRead > 0)), 13);
  *x86_out = (uint32_t)e->Valid | (e->Context << 1) | ((uint32_t)s << 9);
  return nout;
}
int p8f_exe_debug(Exe* e, uint32_t* out) {  /* parser internals for test diagnostics */
  out[0] = e->BrkPoint; out[1] = (uint32_t)e->State; out[2] = e->Op.Data; out[3] = e->Op.Code; out[4] = e->Op.Flags; out[5] = e->Op.BytesRead;
  out[6] = (uint32_t)e->pState; out[7] = e->OpCategMask;
  return 0;
}
