// symbols.h — the two alphabets of the reference, same names and member order.
//   silo::Nucleotide  include/silo/common/nucleotide_symbols.h:12-93, src/silo/common/nucleotide_symbols.cpp:7-85
//   silo::AminoAcid   include/silo/common/aa_symbols.h:12-98,        src/silo/common/aa_symbols.cpp:5-117
// Symbol values are the reference's enum values; they double as the symbol ids of the C ABI
// (include/silo_gpu.h).
#pragma once

#include <array>
#include <cstdint>
#include <optional>
#include <string_view>

#include "silo_gpu.h"

namespace silo {

class Nucleotide {
  public:
   enum class Symbol : uint8_t { GAP, A, C, G, T, R, Y, S, W, K, M, B, D, H, V, N };

   static constexpr uint32_t COUNT = 16;
   static constexpr uint32_t ABI_ALPHABET = SILO_GPU_ALPHABET_NUCLEOTIDE;
   static constexpr std::string_view SYMBOL_NAME = "Nucleotide";
   static constexpr std::string_view SYMBOL_NAME_LOWER_CASE = "nucleotide";

   static constexpr std::array<Symbol, COUNT> SYMBOLS{
      Symbol::GAP, Symbol::A, Symbol::C, Symbol::G, Symbol::T, Symbol::R, Symbol::Y, Symbol::S,
      Symbol::W,   Symbol::K, Symbol::M, Symbol::B, Symbol::D, Symbol::H, Symbol::V, Symbol::N,
   };
   static constexpr std::array<Symbol, 5> VALID_MUTATION_SYMBOLS{
      Symbol::GAP, Symbol::A, Symbol::C, Symbol::G, Symbol::T,
   };
   static constexpr Symbol SYMBOL_MISSING = Symbol::N;

   static char symbolToChar(Symbol symbol) { return "-ACGTRYSWKMBDHVN"[static_cast<uint8_t>(symbol)]; }

   static std::optional<Symbol> charToSymbol(char character) {
      switch (character) {
         case '.':
         case '-': return Symbol::GAP;
         case 'A': return Symbol::A;
         case 'C': return Symbol::C;
         case 'G': return Symbol::G;
         case 'T':
         case 'U': return Symbol::T;
         case 'R': return Symbol::R;
         case 'Y': return Symbol::Y;
         case 'S': return Symbol::S;
         case 'W': return Symbol::W;
         case 'K': return Symbol::K;
         case 'M': return Symbol::M;
         case 'B': return Symbol::B;
         case 'D': return Symbol::D;
         case 'H': return Symbol::H;
         case 'V': return Symbol::V;
         case 'N': return Symbol::N;
         default: return std::nullopt;
      }
   }
};

class AminoAcid {
  public:
   enum class Symbol : uint8_t {
      GAP, A, C, D, E, F, G, H, I, K, L, M, N, P, Q, R, S, T, V, W, Y, B, Z, STOP, X,
   };

   static constexpr uint32_t COUNT = 25;
   static constexpr uint32_t ABI_ALPHABET = SILO_GPU_ALPHABET_AMINO_ACID;
   static constexpr std::string_view SYMBOL_NAME = "Amino Acid";
   static constexpr std::string_view SYMBOL_NAME_LOWER_CASE = "amino acid";

   // iteration order: X before STOP (aa_symbols.h:49-54)
   static constexpr std::array<Symbol, COUNT> SYMBOLS{
      Symbol::GAP, Symbol::A, Symbol::C, Symbol::D, Symbol::E, Symbol::F, Symbol::G,
      Symbol::H,   Symbol::I, Symbol::K, Symbol::L, Symbol::M, Symbol::N, Symbol::P,
      Symbol::Q,   Symbol::R, Symbol::S, Symbol::T, Symbol::V, Symbol::W, Symbol::Y,
      Symbol::B,   Symbol::Z, Symbol::X, Symbol::STOP,
   };
   static constexpr std::array<Symbol, 22> VALID_MUTATION_SYMBOLS{
      Symbol::GAP, Symbol::A, Symbol::C, Symbol::D, Symbol::E, Symbol::F, Symbol::G, Symbol::H,
      Symbol::I,   Symbol::K, Symbol::L, Symbol::M, Symbol::N, Symbol::P, Symbol::Q, Symbol::R,
      Symbol::S,   Symbol::T, Symbol::V, Symbol::W, Symbol::Y, Symbol::STOP,
   };
   static constexpr Symbol SYMBOL_MISSING = Symbol::X;

   static char symbolToChar(Symbol symbol) { return "-ACDEFGHIKLMNPQRSTVWYBZ*X"[static_cast<uint8_t>(symbol)]; }

   static std::optional<Symbol> charToSymbol(char character) {
      constexpr std::string_view chars = "-ACDEFGHIKLMNPQRSTVWYBZ*X";
      const auto index = chars.find(character);
      if (character == '\0' || index == std::string_view::npos) {
         return std::nullopt;
      }
      return static_cast<Symbol>(index);
   }
};

}  // namespace silo
