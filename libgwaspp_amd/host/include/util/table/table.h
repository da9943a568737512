// util/table/table.h -- Table<D> contract of the reference
// (src/util/table/table.h:49-74): dimensions come from two borrowed indexers,
// the table owns `data`.
#pragma once
#include <string>

#include "util/index_set/indexer.h"

namespace util {

template <class D>
class Table {
public:
    Table(indexer *r, indexer *c)
        : rows(r), columns(c), max_row(r->included_size()), max_column(c->included_size()), data(nullptr) {}
    virtual D operator()(int r, int c) { return data[(long)r * max_column + c]; }
    int row_size() const { return max_row; }
    int column_size() const { return max_column; }
    std::string rowID(int i) const { return rows->getIDAtOrderedIndex(i); }
    std::string columnID(int i) const { return columns->getIDAtOrderedIndex(i); }
    const indexer *getRowSet() { return rows; }
    const indexer *getColumnSet() { return columns; }
    virtual ~Table() { delete[] data; }

protected:
    virtual void initialize() = 0;
    indexer *rows, *columns;
    int max_row, max_column;
    D *data;
};

}  // namespace util
