// json.h — minimal JSON value, parser and serialiser for the query surface.
//
// The reference uses nlohmann_json 3.11 (conanfile.py:14), which is not available here; only the
// slice of its behaviour the query path depends on is provided: objects keep keys sorted
// (std::map, like nlohmann's default object_t), numbers remember whether they were written as an
// unsigned / signed integer or a float (is_number_unsigned in the from_json checks), and doubles are
// printed with the shortest round-trip representation.
#pragma once

#include <charconv>
#include <cmath>
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <string_view>
#include <vector>

namespace silo::json {

class ParseError : public std::runtime_error {
  public:
   using std::runtime_error::runtime_error;
};

class Value;

/// Object members in key order (what nlohmann's std::map-based object_t gives), held in one flat vector: a query
/// object has a handful of keys, and a map node per member was most of the parse time.
class ObjectMembers {
  public:
   using Member = std::pair<std::string, Value>;
   ObjectMembers();
   ObjectMembers(const ObjectMembers&);
   ObjectMembers(ObjectMembers&&) noexcept;
   ObjectMembers& operator=(ObjectMembers);
   ~ObjectMembers();
   [[nodiscard]] const Value* find(std::string_view key) const;
   /// Inserts or overwrites (later duplicates win, as in nlohmann).
   Value& insertOrAssign(std::string key, Value value);
   [[nodiscard]] const Member* begin() const;
   [[nodiscard]] const Member* end() const;
   [[nodiscard]] size_t size() const;

  private:
   std::vector<Member>* members_;  // pointer: Value is incomplete here
};

class Value {
  public:
   enum class Kind { Null, Bool, Unsigned, Integer, Float, String, Array, Object };
   using Array = std::vector<Value>;
   using Object = ObjectMembers;

   Value() = default;
   Value(std::nullptr_t) {}
   Value(bool value) : kind_(Kind::Bool), bool_(value) {}
   Value(int32_t value) : kind_(value < 0 ? Kind::Integer : Kind::Unsigned), int_(value), uint_(static_cast<uint64_t>(value)) {}
   Value(int64_t value) : kind_(value < 0 ? Kind::Integer : Kind::Unsigned), int_(value), uint_(static_cast<uint64_t>(value)) {}
   Value(uint64_t value) : kind_(Kind::Unsigned), int_(static_cast<int64_t>(value)), uint_(value) {}
   Value(double value) : kind_(Kind::Float), double_(value) {}
   Value(const char* value) : kind_(Kind::String), string_(value) {}
   Value(std::string value) : kind_(Kind::String), string_(std::move(value)) {}
   Value(Array value) : kind_(Kind::Array), array_(std::make_shared<Array>(std::move(value))) {}
   Value(Object value) : kind_(Kind::Object), object_(std::make_shared<Object>(std::move(value))) {}

   static Value array() { return Value(Array{}); }
   static Value object() { return Value(Object{}); }

   [[nodiscard]] Kind kind() const { return kind_; }
   [[nodiscard]] bool is_null() const { return kind_ == Kind::Null; }
   [[nodiscard]] bool is_boolean() const { return kind_ == Kind::Bool; }
   [[nodiscard]] bool is_number() const { return kind_ == Kind::Unsigned || kind_ == Kind::Integer || kind_ == Kind::Float; }
   [[nodiscard]] bool is_number_unsigned() const { return kind_ == Kind::Unsigned; }
   [[nodiscard]] bool is_number_integer() const { return kind_ == Kind::Unsigned || kind_ == Kind::Integer; }
   [[nodiscard]] bool is_number_float() const { return kind_ == Kind::Float; }
   [[nodiscard]] bool is_string() const { return kind_ == Kind::String; }
   [[nodiscard]] bool is_array() const { return kind_ == Kind::Array; }
   [[nodiscard]] bool is_object() const { return kind_ == Kind::Object; }

   // keys are looked up as views: a lookup by literal builds no std::string (most keys of the query surface are longer
   // than the small-string buffer)
   [[nodiscard]] bool contains(std::string_view key) const {
      return kind_ == Kind::Object && object_->find(key) != nullptr;
   }
   [[nodiscard]] const Value& at(std::string_view key) const {
      if (kind_ != Kind::Object) {
         throw std::out_of_range("json value is not an object");
      }
      const Value* found = object_->find(key);
      if (found == nullptr) {
         throw std::out_of_range("key '" + std::string(key) + "' not found");
      }
      return *found;
   }
   [[nodiscard]] const Value& operator[](std::string_view key) const { return at(key); }
   [[nodiscard]] const Value& operator[](const char* key) const { return at(std::string_view(key)); }
   Value& set(const std::string& key, Value value) {
      if (kind_ != Kind::Object) {
         *this = object();
      }
      return object_->insertOrAssign(key, std::move(value));
   }
   void push_back(Value value) {
      if (kind_ != Kind::Array) {
         *this = array();
      }
      array_->push_back(std::move(value));
   }
   [[nodiscard]] const Array& items() const {
      static const Array empty;
      return kind_ == Kind::Array ? *array_ : empty;
   }
   [[nodiscard]] const Object& members() const {
      static const Object empty;
      return kind_ == Kind::Object ? *object_ : empty;
   }

   [[nodiscard]] bool as_bool() const { return bool_; }
   [[nodiscard]] const std::string& as_string() const { return string_; }
   [[nodiscard]] uint32_t as_uint32() const { return static_cast<uint32_t>(kind_ == Kind::Float ? static_cast<int64_t>(double_) : int_); }
   [[nodiscard]] int64_t as_int64() const { return kind_ == Kind::Float ? static_cast<int64_t>(double_) : int_; }
   [[nodiscard]] double as_double() const {
      if (kind_ == Kind::Float) {
         return double_;
      }
      return kind_ == Kind::Unsigned ? static_cast<double>(uint_) : static_cast<double>(int_);
   }

   [[nodiscard]] std::string dump() const {
      std::string out;
      write(out);
      return out;
   }

   /// Serialisation primitives, also used by writers that do not build a Value tree first.
   static void appendString(std::string& out, const std::string& text) { writeString(out, text); }
   static void appendDouble(std::string& out, double value) {
      if (!std::isfinite(value)) {
         out += "null";  // nlohmann dumps non-finite numbers as null
         return;
      }
      char buffer[40];
      const auto result = std::to_chars(buffer, buffer + sizeof(buffer), value);
      bool integral = true;
      for (const char* c = buffer; c != result.ptr; ++c) {
         integral = integral && *c != '.' && *c != 'e' && *c != 'E';
      }
      out.append(buffer, result.ptr);
      if (integral) {
         out += ".0";
      }
   }

  private:
   Kind kind_ = Kind::Null;
   bool bool_ = false;
   int64_t int_ = 0;
   uint64_t uint_ = 0;
   double double_ = 0.0;
   std::string string_;
   std::shared_ptr<Array> array_;
   std::shared_ptr<Object> object_;

   static void writeString(std::string& out, const std::string& text) {
      out.push_back('"');
      for (const unsigned char c : text) {
         switch (c) {
            case '"': out += "\\\""; break;
            case '\\': out += "\\\\"; break;
            case '\b': out += "\\b"; break;
            case '\f': out += "\\f"; break;
            case '\n': out += "\\n"; break;
            case '\r': out += "\\r"; break;
            case '\t': out += "\\t"; break;
            default:
               if (c < 0x20) {
                  char buffer[8];
                  snprintf(buffer, sizeof(buffer), "\\u%04x", c);
                  out += buffer;
               } else {
                  out.push_back(static_cast<char>(c));
               }
         }
      }
      out.push_back('"');
   }

   void write(std::string& out) const {
      switch (kind_) {
         case Kind::Null: out += "null"; break;
         case Kind::Bool: out += bool_ ? "true" : "false"; break;
         case Kind::Unsigned: out += std::to_string(uint_); break;
         case Kind::Integer: out += std::to_string(int_); break;
         case Kind::Float: {
            appendDouble(out, double_);
            break;
         }
         case Kind::String: writeString(out, string_); break;
         case Kind::Array: {
            out.push_back('[');
            bool first = true;
            for (const Value& item : *array_) {
               if (!first) {
                  out.push_back(',');
               }
               first = false;
               item.write(out);
            }
            out.push_back(']');
            break;
         }
         case Kind::Object: {
            out.push_back('{');
            bool first = true;
            for (const auto& [key, value] : *object_) {
               if (!first) {
                  out.push_back(',');
               }
               first = false;
               writeString(out, key);
               out.push_back(':');
               value.write(out);
            }
            out.push_back('}');
            break;
         }
      }
   }
};

inline ObjectMembers::ObjectMembers() : members_(new std::vector<Member>()) {}
inline ObjectMembers::ObjectMembers(const ObjectMembers& other) : members_(new std::vector<Member>(*other.members_)) {}
inline ObjectMembers::ObjectMembers(ObjectMembers&& other) noexcept : members_(other.members_) {
   other.members_ = nullptr;
}
inline ObjectMembers& ObjectMembers::operator=(ObjectMembers other) {
   std::swap(members_, other.members_);
   return *this;
}
inline ObjectMembers::~ObjectMembers() {
   delete members_;
}
inline const Value* ObjectMembers::find(std::string_view key) const {
   for (const Member& member : *members_) {
      if (member.first == key) {
         return &member.second;
      }
   }
   return nullptr;
}
inline Value& ObjectMembers::insertOrAssign(std::string key, Value value) {
   auto position = members_->begin();
   while (position != members_->end() && position->first < key) {
      ++position;
   }
   if (position != members_->end() && position->first == key) {
      position->second = std::move(value);
      return position->second;
   }
   if (members_->capacity() == 0) {
      const auto offset = position - members_->begin();
      members_->reserve(4);  // a query object has a handful of keys: no regrowth (moving 150-byte members) on the way there
      position = members_->begin() + offset;
   }
   return members_->emplace(position, std::move(key), std::move(value))->second;
}
inline const ObjectMembers::Member* ObjectMembers::begin() const {
   return members_->data();
}
inline const ObjectMembers::Member* ObjectMembers::end() const {
   return members_->data() + members_->size();
}
inline size_t ObjectMembers::size() const {
   return members_->size();
}

class Parser {
  public:
   explicit Parser(const std::string& text) : text_(text) {}

   Value parseDocument() {
      skipWhitespace();
      Value value = parseValue(0);
      skipWhitespace();
      if (pos_ != text_.size()) {
         fail("unexpected trailing characters");
      }
      return value;
   }

  private:
   const std::string& text_;
   size_t pos_ = 0;

   [[noreturn]] void fail(const std::string& what) const {
      throw ParseError("parse error at byte " + std::to_string(pos_ + 1) + ": " + what);
   }
   void skipWhitespace() {
      while (pos_ < text_.size() && (text_[pos_] == ' ' || text_[pos_] == '\t' || text_[pos_] == '\n' || text_[pos_] == '\r')) {
         ++pos_;
      }
   }
   bool consume(const char* literal) {
      const size_t length = std::char_traits<char>::length(literal);
      if (text_.compare(pos_, length, literal) == 0) {
         pos_ += length;
         return true;
      }
      return false;
   }

   Value parseValue(int depth) {
      if (depth > 512) {
         fail("nesting too deep");
      }
      if (pos_ >= text_.size()) {
         fail("unexpected end of input");
      }
      const char c = text_[pos_];
      if (c == '{') {
         return parseObject(depth);
      }
      if (c == '[') {
         return parseArray(depth);
      }
      if (c == '"') {
         return Value(parseString());
      }
      if (consume("true")) {
         return Value(true);
      }
      if (consume("false")) {
         return Value(false);
      }
      if (consume("null")) {
         return Value(nullptr);
      }
      if (c == '-' || (c >= '0' && c <= '9')) {
         return parseNumber();
      }
      fail("syntax error while parsing value - invalid literal");
   }

   Value parseObject(int depth) {
      ++pos_;
      Value::Object object;
      skipWhitespace();
      if (pos_ < text_.size() && text_[pos_] == '}') {
         ++pos_;
         return Value(std::move(object));
      }
      while (true) {
         skipWhitespace();
         if (pos_ >= text_.size() || text_[pos_] != '"') {
            fail("syntax error while parsing object key - expected string literal");
         }
         std::string key = parseString();
         skipWhitespace();
         if (pos_ >= text_.size() || text_[pos_] != ':') {
            fail("syntax error while parsing object separator - expected ':'");
         }
         ++pos_;
         skipWhitespace();
         object.insertOrAssign(std::move(key), parseValue(depth + 1));  // later duplicates win, as in nlohmann
         skipWhitespace();
         if (pos_ < text_.size() && text_[pos_] == ',') {
            ++pos_;
            continue;
         }
         if (pos_ < text_.size() && text_[pos_] == '}') {
            ++pos_;
            return Value(std::move(object));
         }
         fail("syntax error while parsing object - expected ',' or '}'");
      }
   }

   Value parseArray(int depth) {
      ++pos_;
      Value::Array array;
      array.reserve(8);
      skipWhitespace();
      if (pos_ < text_.size() && text_[pos_] == ']') {
         ++pos_;
         return Value(std::move(array));
      }
      while (true) {
         skipWhitespace();
         array.push_back(parseValue(depth + 1));
         skipWhitespace();
         if (pos_ < text_.size() && text_[pos_] == ',') {
            ++pos_;
            continue;
         }
         if (pos_ < text_.size() && text_[pos_] == ']') {
            ++pos_;
            return Value(std::move(array));
         }
         fail("syntax error while parsing array - expected ',' or ']'");
      }
   }

   static void appendUtf8(std::string& out, uint32_t code) {
      if (code < 0x80) {
         out.push_back(static_cast<char>(code));
      } else if (code < 0x800) {
         out.push_back(static_cast<char>(0xC0 | (code >> 6)));
         out.push_back(static_cast<char>(0x80 | (code & 0x3F)));
      } else if (code < 0x10000) {
         out.push_back(static_cast<char>(0xE0 | (code >> 12)));
         out.push_back(static_cast<char>(0x80 | ((code >> 6) & 0x3F)));
         out.push_back(static_cast<char>(0x80 | (code & 0x3F)));
      } else {
         out.push_back(static_cast<char>(0xF0 | (code >> 18)));
         out.push_back(static_cast<char>(0x80 | ((code >> 12) & 0x3F)));
         out.push_back(static_cast<char>(0x80 | ((code >> 6) & 0x3F)));
         out.push_back(static_cast<char>(0x80 | (code & 0x3F)));
      }
   }

   uint32_t parseHex4() {
      if (pos_ + 4 > text_.size()) {
         fail("invalid \\u escape");
      }
      uint32_t code = 0;
      for (int i = 0; i < 4; ++i) {
         const char c = text_[pos_++];
         code <<= 4;
         if (c >= '0' && c <= '9') {
            code |= static_cast<uint32_t>(c - '0');
         } else if (c >= 'a' && c <= 'f') {
            code |= static_cast<uint32_t>(c - 'a' + 10);
         } else if (c >= 'A' && c <= 'F') {
            code |= static_cast<uint32_t>(c - 'A' + 10);
         } else {
            fail("invalid \\u escape");
         }
      }
      return code;
   }

   std::string parseString() {
      ++pos_;
      // the usual string has no escapes: find its end and copy it in one piece
      size_t end = pos_;
      while (end < text_.size() && text_[end] != '"' && text_[end] != '\\' && static_cast<unsigned char>(text_[end]) >= 0x20) {
         ++end;
      }
      if (end < text_.size() && text_[end] == '"') {
         std::string whole(text_.data() + pos_, end - pos_);
         pos_ = end + 1;
         return whole;
      }
      std::string out;
      while (true) {
         if (pos_ >= text_.size()) {
            fail("syntax error while parsing string - missing closing quote");
         }
         const char c = text_[pos_++];
         if (c == '"') {
            return out;
         }
         if (static_cast<unsigned char>(c) < 0x20) {
            fail("syntax error while parsing string - control character must be escaped");
         }
         if (c != '\\') {
            out.push_back(c);
            continue;
         }
         if (pos_ >= text_.size()) {
            fail("invalid escape");
         }
         const char e = text_[pos_++];
         switch (e) {
            case '"': out.push_back('"'); break;
            case '\\': out.push_back('\\'); break;
            case '/': out.push_back('/'); break;
            case 'b': out.push_back('\b'); break;
            case 'f': out.push_back('\f'); break;
            case 'n': out.push_back('\n'); break;
            case 'r': out.push_back('\r'); break;
            case 't': out.push_back('\t'); break;
            case 'u': {
               uint32_t code = parseHex4();
               if (code >= 0xD800 && code <= 0xDBFF && pos_ + 1 < text_.size() && text_[pos_] == '\\' && text_[pos_ + 1] == 'u') {
                  pos_ += 2;
                  const uint32_t low = parseHex4();
                  code = 0x10000 + ((code - 0xD800) << 10) + (low - 0xDC00);
               }
               appendUtf8(out, code);
               break;
            }
            default: fail("invalid escape");
         }
      }
   }

   Value parseNumber() {
      const size_t start = pos_;
      bool negative = false;
      bool is_float = false;
      if (text_[pos_] == '-') {
         negative = true;
         ++pos_;
      }
      if (pos_ >= text_.size() || text_[pos_] < '0' || text_[pos_] > '9') {
         fail("syntax error while parsing value - invalid number; expected digit after '-'");
      }
      if (text_[pos_] == '0') {
         ++pos_;
      } else {
         while (pos_ < text_.size() && text_[pos_] >= '0' && text_[pos_] <= '9') {
            ++pos_;
         }
      }
      if (pos_ < text_.size() && text_[pos_] == '.') {
         is_float = true;
         ++pos_;
         if (pos_ >= text_.size() || text_[pos_] < '0' || text_[pos_] > '9') {
            fail("syntax error while parsing value - invalid number; expected digit after '.'");
         }
         while (pos_ < text_.size() && text_[pos_] >= '0' && text_[pos_] <= '9') {
            ++pos_;
         }
      }
      if (pos_ < text_.size() && (text_[pos_] == 'e' || text_[pos_] == 'E')) {
         is_float = true;
         ++pos_;
         if (pos_ < text_.size() && (text_[pos_] == '+' || text_[pos_] == '-')) {
            ++pos_;
         }
         if (pos_ >= text_.size() || text_[pos_] < '0' || text_[pos_] > '9') {
            fail("syntax error while parsing value - invalid number; expected digit in exponent");
         }
         while (pos_ < text_.size() && text_[pos_] >= '0' && text_[pos_] <= '9') {
            ++pos_;
         }
      }
      const char* begin = text_.data() + start;
      const char* end = text_.data() + pos_;
      if (!is_float) {
         if (negative) {
            int64_t value = 0;
            const auto result = std::from_chars(begin, end, value);
            if (result.ec == std::errc() && result.ptr == end) {
               return Value(value);
            }
         } else {
            uint64_t value = 0;
            const auto result = std::from_chars(begin, end, value);
            if (result.ec == std::errc() && result.ptr == end) {
               return Value(value);
            }
         }
      }
      double value = 0.0;
      const auto result = std::from_chars(begin, end, value);
      if (result.ec != std::errc() && result.ec != std::errc::result_out_of_range) {
         fail("invalid number");
      }
      return Value(value);
   }
};

inline Value parse(const std::string& text) {
   return Parser(text).parseDocument();
}

}  // namespace silo::json
